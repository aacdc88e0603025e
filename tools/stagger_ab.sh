for st in 0 1 2 3 4 0 2; do echo "stagger=$st"; RMR_FUSED_STAGGER=$st python tools/ab_variants.py --libs default --dtype bf16 --cfgs C100 2>&1 | cut -c1-120; done
