# fp32-storage head convs: exact fp32 MFMA vs bf16x3 (tools/kbench2.py / kbench.py)
python tools/kbench2.py --f32 f32 --variants 0 --reps 10
python tools/kbench2.py --f32 bf16x3 --variants 0 --reps 10
for a in f32 bf16x3; do for c in "256 256" "64 256" "256 720"; do set -- $c; echo "wgrad $a $1->$2"; python tools/kbench.py --dtype f32 --arith $a --which wgrad --cin $1 --cout $2 --reps 10; done; done
