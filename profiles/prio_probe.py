import torch
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
for p in (-2,-1,0,1,2):
    try:
        s = torch.cuda.Stream(priority=p); print(p, "ok", s.priority)
    except Exception as e:
        print(p, "err", e)
