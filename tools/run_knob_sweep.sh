#!/bin/bash
# two rounds over the step's scheduling knobs on one box (bench.py --steps 40): side-stream threshold, 256x256-kernel threshold,
# fused-stage-kernel K limit, ahead-of-step AdamW.  Last run (final round-2 build): everything within the +-0.1 ms run-to-run
# noise of the defaults (12.38-12.51 ms) except SGV_EARLY_ADAM=0 (12.67-12.76).
b() { env "$@" python bench.py --steps 40 --warmup 5 --cpu-baseline skip --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for r in 1 2; do
echo "base $(b A=1)"
echo "side_maxgf=100 $(b SGV_DW_SIDE_MAXGF=100)"
echo "side_maxgf=450 $(b SGV_DW_SIDE_MAXGF=450)"
echo "side_maxgf=700 $(b SGV_DW_SIDE_MAXGF=700)"
echo "t256_min_gf=15 $(b SGV_T256_MIN_GF=15)"
echo "t256_min_gf=60 $(b SGV_T256_MIN_GF=60)"
echo "convgn_maxk=2048 $(b SGV_CONVGN_MAXK=2048)"
echo "convgn_maxk=8192 $(b SGV_CONVGN_MAXK=8192)"
echo "early_adam=0 $(b SGV_EARLY_ADAM=0)"
done
