set -o pipefail
out=gpurun_out/r04r
mkdir -p $out


P=tools/dev/asm_probes
bash tools/dev/ab_hsaco_b.sh $out/ab base=$P/b_base.hsaco ud12=$P/b_ud12.hsaco ud6=$P/b_ud6.hsaco late=$P/b_dma_late.hsaco sp2=$P/b_dma_sp2.hsaco sp3=$P/b_dma_sp3.hsaco
