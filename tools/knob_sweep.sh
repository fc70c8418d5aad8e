# usage (GPU box): bash tools/knob_sweep.sh -- development: the bench step under the library's tuning switches, one at a time
# (60-step lines; the defaults are printed three times to show the run-to-run spread)
S="|--steps 60"
bash tools/ab_bench.sh "default=$S" \
  "bn_gy64=MVK_BN_FUSED_GY=64$S" "bn_gy256=MVK_BN_FUSED_GY=256$S" \
  "bn_mid512=MVK_BN_MID_ROWS=512$S" "bn_mid2048=MVK_BN_MID_ROWS=2048$S" "bn_mid4096=MVK_BN_MID_ROWS=4096$S" \
  "default=$S" \
  "bn_small64=MVK_BN_SMALL_ROWS=64$S" "bn_small256=MVK_BN_SMALL_ROWS=256$S" \
  "stats_bonus0=MVK_GEMM_STATS_BONUS=0$S" "stats_bonus30k=MVK_GEMM_STATS_BONUS=30000$S" \
  "dw_kt16=MVK_DW_GROUP_KTILES=16$S" "dw_kt32=MVK_DW_GROUP_KTILES=32$S" "dw_kt64=MVK_DW_GROUP_KTILES=64$S" "dw_kt96=MVK_DW_GROUP_KTILES=96$S" \
  "default=$S" \
  "gather_split0=MVK_GATHER_SPLIT=0$S" "scatter_spread0=MVK_SCATTER_SPREAD=0$S" "sub2048=MVK_SUB_MULTI_MIN=2048 MVK_NB_MULTI_MIN=1024$S" \
  "stats_rows16k=MVK_GEMM_STATS_MAX_ROWS=16384$S" "stats_rows64k=MVK_GEMM_STATS_MAX_ROWS=65536$S"
