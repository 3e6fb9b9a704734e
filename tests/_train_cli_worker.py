"""Worker of tests/test_gpu_cli.py::test_train_cli_on_two_ranks: train.main() under torchrun (the ranks share the GPU through
gloo); every rank leaves its final parameters in <model>/rank<r>.pt for the test to compare."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hair-gs_amd")]
import torch

import train as train_cli

src, model = sys.argv[1], sys.argv[2]
rank = int(os.environ.get("RANK", "0"))
scene = train_cli.main(["-s", src, "-m", model, "--iterations", "24", "--save_frequency", "12", "--quiet", "--densify_from_iter", "2",
                        "--densification_interval", "5", "--opacity_reset_interval", "11", "--densify_grad_threshold", "1e-7"])
g = scene.gaussians
torch.save({"xyz": g._xyz.detach().cpu(), "opacity": g._opacity.detach().cpu(), "scaling": g._scaling.detach().cpu(),
            "dc": g._features_dc.detach().cpu()}, os.path.join(model, f"rank{rank}.pt"))
print(f"TRAIN_CLI_RANK_{rank}_OK P={g._xyz.shape[0]}")
