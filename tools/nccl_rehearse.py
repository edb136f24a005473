"""World-size-1 RCCL group: exercises the overlapped exchange (async all_reduce(AVG) of arena ranges on the
RCCL stream between backward launches) against the real RCCL stream machinery on one GPU."""
import os, sys, time
if os.environ.get('NO_QUEUE_FIX') != '1':
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')       # as bench.py / the package do (see their comments)
import torch, torch.distributed as dist
sys.path.insert(0, '/root/repo')
import objectdetection.pytorch_amd      # sets GPU_MAX_HW_QUEUES too if still unset: after `import torch`, before any GPU call
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
os.environ['MI355_FORCE_OVERLAP'] = '1'
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
if os.environ.get('NO_PG') != '1':
    dist.init_process_group('nccl', rank=0, world_size=1, **({'device_id': dev} if os.environ.get('NO_DEVICE_ID') != '1' else {}))
import bench, synth
from objectdetection.pytorch_amd.trainer import Trainer
model, sd = bench.build_model(50, 80, dev)
img = torch.from_numpy(synth.synth_images(16, 512, 512, seed=1)).to(dev)
ann = torch.from_numpy(synth.synth_annotations(16, 512, 512, 8, 16, 80, seed=2)).to(dev)
seg = int(os.environ.get('SEG_MB', '16')) << 20
tr = Trainer(model, lr=1e-4, weight_decay=5e-4, clip_norm=0.1, max_annots=16, segment_bytes=seg,
             overlap=os.environ.get('MI355_DP_OVERLAP', '1') != '0')
for _ in range(5): l = tr.step(img, ann)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): l = tr.step(img, ann)
torch.cuda.synchronize()
plan = model._plans[next(iter(model._plans))]
print('hw_queues=%s two_streams=%s overlap=%s seg=%s MB: %.2f ms/step, loss %s' % (os.environ.get('GPU_MAX_HW_QUEUES'), bool(plan.use_side), tr._overlap(), os.environ.get('SEG_MB', '16'), (time.perf_counter() - t0) / 20 * 1e3, l.tolist()), flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
