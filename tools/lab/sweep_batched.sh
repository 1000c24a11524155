P="python tools/batch_decode_probe.py --steps 16"
echo "== qmm3 min rows"; for m in 9 4 2; do for b in 2 4 8; do echo -n "min_m=$m "; TL_QMM3_MIN_M=$m $P --batch $b; done; done
echo "== attention at batch 64"; for cfg in "TL_ATTN_RQ=4 TL_ATTN_MIN_TOKENS=64" "TL_ATTN_RQ=4 TL_ATTN_MIN_TOKENS=256" "TL_ATTN_RQ=4 TL_ATTN_MIN_TOKENS=512" "TL_ATTN_RQ=1 TL_ATTN_MIN_TOKENS=64" "TL_ATTN_RQ=1 TL_ATTN_MIN_TOKENS=256" "TL_ATTN_RQ=1 TL_ATTN_MIN_TOKENS=512"; do echo -n "$cfg "; env $cfg $P --batch 64; done
echo "== attention at batch 16"; for cfg in "TL_ATTN_RQ=4 TL_ATTN_MIN_TOKENS=64" "TL_ATTN_RQ=4 TL_ATTN_MIN_TOKENS=256" "TL_ATTN_RQ=1 TL_ATTN_MIN_TOKENS=64" "TL_ATTN_RQ=1 TL_ATTN_MIN_TOKENS=256"; do echo -n "$cfg "; env $cfg $P --batch 16; done
