"""sha256 (first 16 hex digits) over the kernel sources (csrc/*.hip, *.h, file names included): a committed PMC traffic
figure belongs to the kernels it was collected on (bench.py `roofline.traffic_stale`, tools/collect_traffic.py)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'objectdetection', 'pytorch_amd', 'csrc')
    for fn in sorted(os.listdir(d)):
        if fn.endswith(('.hip', '.h')):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), 'rb').read())
    return h.hexdigest()[:16]


if __name__ == '__main__':
    print(csrc_hash())
