"""What does torch / RCCL on this ROCm build allow inside a hipGraph capture?  One rank, one GPU.  Each case in its own
try / except; failed graphs are kept alive (destroying one whose stream still captures aborts the process)."""
import os, sys, traceback
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29733')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1)
x = torch.ones(1 << 20, device='cuda')
dist.all_reduce(x); torch.cuda.synchronize()                       # communicator set up eagerly
y = torch.empty_like(x)
dist.all_gather_into_tensor(y, x); torch.cuda.synchronize()
keep = []


def case(name, fn, mode='thread_local'):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph(); keep.append(g)
    try:
        with torch.cuda.stream(s):
            g.capture_begin(capture_error_mode=mode)
            try:
                fn()
            except BaseException:
                print(f'[{name}/{mode}] raised inside capture:\n' + traceback.format_exc(limit=3), flush=True)
                try:
                    g.capture_end()
                except BaseException as e2:
                    print(f'[{name}/{mode}] capture_end after failure: {type(e2).__name__}: {str(e2)[:200]}', flush=True)
                return
            g.capture_end()
        torch.cuda.synchronize()
        x.fill_(1.0); torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        print(f'[{name}/{mode}] captured and replayed; x[0] = {float(x[0])}', flush=True)
    except BaseException:
        print(f'[{name}/{mode}] failed:\n' + traceback.format_exc(limit=3), flush=True)


def sync_ar():
    x.mul_(2.0); dist.all_reduce(x); x.add_(1.0)


def async_ar():
    x.mul_(2.0)
    w = dist.all_reduce(x, async_op=True)
    z = y * 2.0                     # work beside the reduction
    w.wait()
    x.add_(z[: x.numel()] * 0 + 1.0)


def gather():
    dist.all_gather_into_tensor(y, x)


for mode in ('thread_local', 'relaxed', 'global'):
    case('sync all_reduce', sync_ar, mode)
    case('async all_reduce + wait', async_ar, mode)
    case('all_gather_into_tensor', gather, mode)
print('probe done', flush=True)
dist.destroy_process_group()
