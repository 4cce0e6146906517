"""Host time per call of the hot entry point on a tiny tensor: single GPU, forced 1-rank exchange through the collective
(CNNQ_XRANK=0) and through the in-launch exchange (CNNQ_XRANK=1).  Run with CNNQ_FORCE_EXCHANGE=1 for the last two."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
forced = os.environ.get('CNNQ_FORCE_EXCHANGE', '0') == '1'
if forced:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29611')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0), rank=0, world_size=1)
from cnn_quantization_amd import ops
x = torch.randn(2, 8, 4, 4, device='cuda'); y = torch.empty_like(x)
fn = lambda: ops.act_qdq_per_channel(x, 4, out=y, group=None)
for _ in range(200): fn()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3000): fn()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('forced=%s CNNQ_XRANK=%s: host %.1f us per call (gpu drained after %.1f us more per call)' % (forced, os.environ.get('CNNQ_XRANK', 'auto'), (t1 - t0) / 3000 * 1e6, (t2 - t1) / 3000 * 1e6))
if forced:
    from cnn_quantization_amd import distributed as D, rccl
    if D.xrank_exchange(None) is not None: D.xrank_exchange(None).close()
    rccl.close_all(); dist.destroy_process_group()
