set -o pipefail
out=gpurun_out/r04j
mkdir -p $out
P=tools/dev/asm_probes
bash tools/dev/ab_hsaco.sh $out/ab v2=$P/v2.hsaco da=$P/da.hsaco db=$P/db.hsaco dc=$P/dc.hsaco dd=$P/dd.hsaco de=$P/de.hsaco df=$P/df.hsaco dg=$P/dg.hsaco dcf=$P/dcf.hsaco dcl=$P/dcl.hsaco
