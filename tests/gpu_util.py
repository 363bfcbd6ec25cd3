"""Device-memory helpers for the GPU tests: torch is plumbing (allocation, copies), nothing else."""
import numpy as np
import torch


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def dev_empty_f32(n):
    return torch.empty(int(n), dtype=torch.float32, device="cuda")


def ptr(t):
    return t.data_ptr()


def to_host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()
