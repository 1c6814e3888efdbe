#!/bin/bash
# profiles/r06_wgrad_ceiling.txt.  Run on the GPU box from the repo root; the lab builds must exist (build container:
#   bash tools/lab/build_variants.sh wsb_noepi:-DHERO_WSB_NOEPI wsb_dmaonly:-DHERO_WSB_NOMFMA,-DHERO_WSB_NOLDF wsb_mfmaonly:-DHERO_WSB_NOLOADS,-DHERO_WSB_NOLDF)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 600 python tools/lab/wgrad_ceiling.py product gpurun_out/wceil_product.json > gpurun_out/wceil_product.log 2>&1
for v in noepi dmaonly mfmaonly; do
  HERO_HIP_LIB=$PWD/tools/lab/libhero_wsb_$v.so timeout 600 python tools/lab/wgrad_ceiling.py $v gpurun_out/wceil_$v.json > gpurun_out/wceil_$v.log 2>&1
done
python tools/lab/wgrad_ceiling.py table gpurun_out/wceil_product.json gpurun_out/wceil_noepi.json gpurun_out/wceil_dmaonly.json gpurun_out/wceil_mfmaonly.json > gpurun_out/r06_wgrad_ceiling.txt 2>&1
cat gpurun_out/r06_wgrad_ceiling.txt
