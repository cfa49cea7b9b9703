"""top-k selection of commu_sample_topk against torch.topk, with and without a (zero) rejection mask."""
import sys, numpy as np, torch
sys.path.insert(0, "commu-code_amd")
from commu_amd import ops
g = torch.Generator().manual_seed(5)
for trial in range(6):
    logits = torch.randn(64, 729, generator=g) * (1 + trial)
    p = torch.softmax(torch.cat([torch.full((64, 1), float("-inf")), logits[:, 1:] / 0.95], 1), 1)
    ref = p.topk(32, dim=1).indices.sort(1).values
    for use_wrong in (False, True):
        dev = logits.clone().cuda()
        pr = torch.zeros(64, 768, device="cuda")
        w = torch.zeros(64, 729, dtype=torch.uint8, device="cuda") if use_wrong else None
        ops.sample_topk(dev, 0.95, 32, wrong=w, uniforms=torch.full((64,), 0.5, device="cuda"), probs_out=pr)
        got = [(pr[i, :729].cpu() > 0).nonzero().flatten().tolist() for i in range(64)]
        bad = [i for i in range(64) if got[i] != ref[i].tolist()]
        print(trial, use_wrong, "rows differing from torch.topk:", len(bad))
