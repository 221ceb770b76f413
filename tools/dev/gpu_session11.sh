set -o pipefail
out=gpurun_out/r04o
mkdir -p $out /tmp/w4
hipcc -O2 -w -o /tmp/asm_harness tools/dev/asm_harness.cpp || exit 1
G=unet-medical-image-contour-segmentation-cpp_amd/csrc/asm/gen_wino4b_asm.py
L=/opt/rocm/lib/llvm/bin
export ASM_MODE=b
for spec in "1 none" "3 none" "5 lds" "5 vgpr" "8 lds" "8 agpr" "8 vgpr" "9 agpr" "9 vgpr" "0 none"; do
  set -- $spec
  n=$1; what=$2; tag=b${n}_${what}
  if [ "$what" = none ]; then python3 $G /tmp/w4/$tag.s --stop $n 2>/dev/null; else python3 $G /tmp/w4/$tag.s --stop $n --dump $what 2>/dev/null; fi
  $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/w4/$tag.s -o /tmp/w4/$tag.o && $L/ld.lld -shared /tmp/w4/$tag.o -o /tmp/w4/$tag.hsaco || exit 1
  mkdir -p $out/$tag
  timeout -k 5 60 /tmp/asm_harness /tmp/w4/$tag.hsaco 4 16 32 64 64 nopool $out/$tag 2>&1 | tee -a $out/harness.txt
  rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "$tag FAILED rc=$rc"; exit 1; fi
  rm -f $out/$tag/u.bin
  [ "$tag" != "b5_lds" ] && rm -f $out/$tag/in.bin $out/$tag/bias.bin
done
ASM_OP=conv3x3_wino4b timeout -k 10 600 python tools/dev/asm_bringup.py 1,16,32,64,64 1,16,32,64,64,exact 1,32,64,64,64 2,32,64,128,64 1,32,32,64,64,pool 16,128,128,64,64 4,64,64,128,128,pool > $out/asm_bringup.txt 2>&1; rc=$?; grep -v "bad rows\|bad cols\|bad channels" $out/asm_bringup.txt; exit $rc
