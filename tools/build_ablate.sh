#!/bin/bash
# timing-only ablation builds of the LUT table kernel (results are wrong by construction)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ablate
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -Imct_quantizers_amd/csrc"
for v in LDS DIV BOTH STAGE; do
  D=""; [ $v = LDS ] && D="-DMCTQ_ABLATE_LDS"; [ $v = DIV ] && D="-DMCTQ_ABLATE_DIV"; [ $v = BOTH ] && D="-DMCTQ_ABLATE_LDS -DMCTQ_ABLATE_DIV"; [ $v = STAGE ] && D="-DMCTQ_ABLATE_STAGE"
  ( hipcc $F $D -shared -o tools/ablate/libmctq_$v.so mct_quantizers_amd/csrc/mctq_misc.hip mct_quantizers_amd/csrc/mctq_lut_table.hip mct_quantizers_amd/csrc/mctq_affine.hip mct_quantizers_amd/csrc/mctq_lut_scan.hip mct_quantizers_amd/csrc/mctq_codes.hip ) &
done
wait
ls -la tools/ablate
