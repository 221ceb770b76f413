set -o pipefail
mkdir -p gpurun_out/r04d /tmp/w4
hipcc -O2 -w -o /tmp/asm_harness tools/dev/asm_harness.cpp || exit 1
G=unet-medical-image-contour-segmentation-cpp_amd/csrc/asm/gen_wino4_asm.py
L=/opt/rocm/lib/llvm/bin
rm -f gpurun_out/r04d/harness.txt
for spec in "5 lds" "5 vgpr" "7 lds" "8 lds" "8 agpr" "8 vgpr" "9 agpr" "9 vgpr" "0 none"; do
  set -- $spec
  n=$1; what=$2; tag=s${n}_${what}
  if [ "$what" = none ]; then python3 $G /tmp/w4/$tag.s 2>/dev/null; else python3 $G /tmp/w4/$tag.s --stop $n --dump $what 2>/dev/null; fi
  $L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c /tmp/w4/$tag.s -o /tmp/w4/$tag.o && $L/ld.lld -shared /tmp/w4/$tag.o -o /tmp/w4/$tag.hsaco || exit 1
  mkdir -p gpurun_out/r04d/$tag
  timeout -k 5 60 /tmp/asm_harness /tmp/w4/$tag.hsaco 4 16 16 64 128 nopool gpurun_out/r04d/$tag 2>&1 | tee -a gpurun_out/r04d/harness.txt
  rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "$tag FAILED rc=$rc"; exit 1; fi
  rm -f gpurun_out/r04d/$tag/u.bin
  [ "$tag" != "s5_lds" ] && rm -f gpurun_out/r04d/$tag/in.bin gpurun_out/r04d/$tag/bias.bin
done
make -C unet-medical-image-contour-segmentation-cpp_amd -j8 > /dev/null 2>&1 || exit 1
timeout -k 10 600 python tools/dev/asm_bringup.py > gpurun_out/r04d/asm_bringup.txt 2>&1; rc=$?; grep -v "bad rows\|bad cols\|bad channels" gpurun_out/r04d/asm_bringup.txt; exit $rc
