import sys
sys.path.insert(0,"unet-medical-image-contour-segmentation-cpp_amd"); sys.path.insert(0,"tests")
import numpy as np, oracle_lib as orc
from miunet import binding, synth
from miunet.spec import UNetSpec, pack_weights
for spec,hw in [(UNetSpec(1,16,1,3),(32,32)),(UNetSpec(1,64,1,3),(32,32)),(UNetSpec(1,64,2,3),(64,64)),(UNetSpec(),(128,128))]:
    blob=pack_weights(spec,synth.make_weights(spec,4321)); imgs=synth.make_images(2,hw[0],hw[1],1,0xBEEF,"blobs")
    r16,_=orc.unet_forward(blob,imgs,bf16=True); r32,_=orc.unet_forward(blob,imgs)
    with binding.Engine(hw[0],hw[1],1,spec.base,spec.levels,3,max_batch=2,conv_algo="bf16") as e:
        e.load_weights(blob); _,g16=e.infer(imgs,want_logits=True)
    with binding.Engine(hw[0],hw[1],1,spec.base,spec.levels,3,max_batch=2,conv_algo="direct") as e:
        e.load_weights(blob); _,g32=e.infer(imgs,want_logits=True)
    print(spec.base,spec.levels,"g16-r16",np.abs(g16-r16).max(),"g16-r32",np.abs(g16-r32).max(),"g32-r32",np.abs(g32-r32).max(),"r16-r32",np.abs(r16-r32).max(),"g16-g32",np.abs(g16-g32).max())
