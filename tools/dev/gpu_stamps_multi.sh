# usage: gpu_stamps_multi.sh <outdir> "name:flags" ...   -- stamps on two shapes (K = 8 and K = 64) for several generator variants, one call
set -o pipefail
out=$1; shift
mkdir -p $out /tmp/w4
hipcc -O2 -w -o /tmp/asm_harness tools/dev/asm_harness.cpp || exit 1
G=unet-medical-image-contour-segmentation-cpp_amd/csrc/asm/gen_wino4_asm.py
L=/opt/rocm/lib/llvm/bin
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  python3 $G /tmp/w4/$name.s --stamps $flags 2>/dev/null && $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/w4/$name.s -o /tmp/w4/$name.o && $L/ld.lld -shared /tmp/w4/$name.o -o /tmp/w4/$name.hsaco || exit 1
done
for round in 1 2; do
for v in "$@"; do
  name=${v%%:*}
  for shape in "16 256 256 128 128" "16 64 64 1024 512"; do
    echo "== $name ($shape) round $round" | tee -a $out/stamps.txt
    timeout -k 5 120 /tmp/asm_harness /tmp/w4/$name.hsaco $shape stamps 2>&1 | grep -v "completed in" | tee -a $out/stamps.txt || exit 1
  done
done
done
echo "== grid caps (base)" | tee -a $out/stamps.txt
for g in 128 64; do ASM_GRID=$g timeout -k 5 120 /tmp/asm_harness /tmp/w4/base.hsaco 16 256 256 128 128 stamps 2>&1 | grep -v "completed in" | tee -a $out/stamps.txt; done
