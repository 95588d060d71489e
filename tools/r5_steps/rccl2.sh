# can RCCL form a 2-rank group on ONE GPU here?  (if it can, the multi-rank captured DDP path can be validated on this box)
cat > /tmp/rccl2.py <<'PY'
import os, sys, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='env://')
x = torch.full((1 << 20,), float(dist.get_rank() + 1), device='cuda')
dist.all_reduce(x); torch.cuda.synchronize()
print('rank', dist.get_rank(), 'allreduce ->', float(x[0]), flush=True)
dist.barrier(); dist.destroy_process_group()
PY
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 /tmp/rccl2.py > $OUT/rccl2.log 2>&1; echo "rccl2 rc=$?" | tee -a $OUT/rc.txt
grep -v "amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*" $OUT/rccl2.log | tail -12
