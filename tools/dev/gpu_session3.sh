set -o pipefail
mkdir -p gpurun_out/r04c /tmp/w4
hipcc -O2 -o /tmp/asm_harness tools/dev/asm_harness.cpp || exit 1
G=unet-medical-image-contour-segmentation-cpp_amd/csrc/asm/gen_wino4_asm.py
L=/opt/rocm/lib/llvm/bin
for n in 1 2 3 4 5 6 7 8 9 0; do
  python3 $G /tmp/w4/k$n.s --stop $n 2>/dev/null && $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/w4/k$n.s -o /tmp/w4/k$n.o && $L/ld.lld -shared /tmp/w4/k$n.o -o /tmp/w4/k$n.hsaco || exit 1
done
for n in 1 2 3 4 5 6 7 8 9 0; do
  echo "== stop at checkpoint $n"
  timeout -k 5 60 /tmp/asm_harness /tmp/w4/k$n.hsaco 1 16 16 64 128 2>&1 | tee -a gpurun_out/r04c/harness.txt
  rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "checkpoint $n FAILED rc=$rc" | tee -a gpurun_out/r04c/harness.txt; exit 1; fi
done
echo "all checkpoints ran"
timeout -k 10 600 python tools/dev/asm_bringup.py > gpurun_out/r04c/asm_bringup.txt 2>&1; rc=$?; cat gpurun_out/r04c/asm_bringup.txt; exit $rc
