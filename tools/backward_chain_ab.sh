#!/bin/bash
# A/B of the backward-chain walk (tools/backward_chain.py) over the data seeds of the fixtures and three library
# configurations; one process per run (the library reads its TG_* switches once).
# Usage: tools/backward_chain_ab.sh OUTDIR [step|gy]
set -e
out=${1:-gpurun_out/chain_ab}
mode=${2:-step}
mkdir -p "$out"
if [ "$mode" = gy ]; then
  tags="c2_b16_256_gy_s500 c2_b16_256_gy_s501 c2_b16_256_gy_s502 c2_b16_256_gy_s503 c2_b16_256_gy_s504"
else
  tags="c2_b16_256 c2_b16_256_s501 c2_b16_256_s502 c2_b16_256_s503 c2_b16_256_s504"
fi
for tag in $tags; do
  fx=steps_chain_seeds; [ "$tag" = c2_b16_256 ] && fx=steps_chain; [ "$mode" = gy ] && fx=steps_chain_gy
  python tools/backward_chain.py --tag $tag --fixture $fx --out $out/${tag}_default.json > $out/${tag}_default.txt 2>&1
  TG_NO_WINO=1 python tools/backward_chain.py --tag $tag --fixture $fx --out $out/${tag}_nowino.json > $out/${tag}_nowino.txt 2>&1
  TG_NO_BN_SMALL=1 python tools/backward_chain.py --tag $tag --fixture $fx --out $out/${tag}_nobnsmall.json > $out/${tag}_nobnsmall.txt 2>&1
  echo "$tag done"
done
