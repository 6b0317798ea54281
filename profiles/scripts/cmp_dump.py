import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
print(sys.argv[1], sys.argv[2], "identical" if torch.equal(a, b) else f"DIFFER max {float((a-b).abs().max())} n {int((a!=b).sum())} of {a.numel()}", "finite", bool(torch.isfinite(a).all()), float(a.abs().mean()))
