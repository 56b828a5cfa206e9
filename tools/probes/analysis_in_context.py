import sys, os
sys.argv = [sys.argv[0], "46", "6"]
mode = os.environ.get("MODE", "")
if "torch" in mode:
    import torch
    if "cuda" in mode: torch.cuda.init(); x = torch.zeros(8, device="cuda"); torch.cuda.synchronize()
sys.path.insert(0, os.getcwd())
exec(open("tools/analysis_time.py").read())
