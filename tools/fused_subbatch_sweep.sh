for sb in 32768 65536 131072 262144 524288; do echo "RMR_FUSED_SUBBATCH=$sb"; RMR_FUSED_SUBBATCH=$sb python tools/ab_variants.py --libs default --dtype bf16 --n 1048576 2>&1 | cut -c1-120; done
