"""Does a float atomic add issued right behind a hipMemsetAsync of the same buffer lose updates?  (torch.zeros of a large
tensor is a memset; index_add_ uses float atomics.)  Exact integer counts are expected."""
import torch

torch.manual_seed(0)
dev = "cuda"
n = 1 << 19
idx = torch.randint(0, n, (1 << 23,), device=dev)
ones = torch.ones(idx.numel(), device=dev)
ref = torch.bincount(idx, minlength=n).float()
keep = []
for trial in range(12):
    how = ("zeros", "empty+fill_", "empty+zero_")[trial % 3]
    if how == "zeros":
        x = torch.zeros(n, device=dev)
    elif how == "empty+fill_":
        x = torch.empty(n, device=dev); x.fill_(0.0)
    else:
        x = torch.empty(n, device=dev); x.zero_()
    x.index_add_(0, idx, ones)
    bad = int((x != ref).sum())
    print(trial, how, "mismatching elements:", bad, flush=True)
    if trial % 2:
        keep.append(x)          # vary which block the next trial gets
    junk = torch.full((n,), 123.0, device=dev); del junk
