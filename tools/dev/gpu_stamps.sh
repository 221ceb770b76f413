# usage: gpu_stamps.sh <outfile> [generator flags...]   -- per-phase cycle stamps of the assembly kernel on four real layer shapes
set -o pipefail
out=$1; shift
mkdir -p $(dirname $out) /tmp/w4
hipcc -O2 -w -o /tmp/asm_harness tools/dev/asm_harness.cpp || exit 1
G=unet-medical-image-contour-segmentation-cpp_amd/csrc/asm/gen_wino4_asm.py
L=/opt/rocm/lib/llvm/bin
python3 $G /tmp/w4/st.s --stamps "$@" 2>/dev/null && $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/w4/st.s -o /tmp/w4/st.o && $L/ld.lld -shared /tmp/w4/st.o -o /tmp/w4/st.hsaco || exit 1
for shape in "16 256 256 128 128" "16 128 128 256 256" "16 64 64 1024 512" "16 32 32 1024 1024"; do
  timeout -k 5 120 /tmp/asm_harness /tmp/w4/st.hsaco $shape stamps 2>&1 | tee -a $out || exit 1
done
