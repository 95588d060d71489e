for k in 16 8; do export EFFDET_IGEMM_NARROW_K=$k; echo K=$k; bash tools/ab_env.sh EFFDET_IGEMM_NARROW 64 128 256; done
export EFFDET_IGEMM_NARROW_K=16; bash tools/ab_env.sh EFFDET_IGEMM_NARROW 0 64
