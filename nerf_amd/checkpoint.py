"""Checkpoint and evaluator interop with the reference.

The reference writes ``{'net': net.state_dict(), 'optim', 'scheduler', 'recorder', 'epoch'}`` with ``torch.save``
(src/utils/net_utils.py:325-340) into ``<trained_model_dir>/<epoch>.pth`` / ``latest.pth`` and reads it back in
``load_network`` (:348-381, newest = ``latest.pth`` if present else the largest epoch number).  ``Network`` /
``KiloNetwork`` of this package have the reference's state-dict keys and shapes
(tests/golden/state_dict_layout.json, generated from the real reference), so a trained lego checkpoint loads
directly.  Files are read with ``torch.load(..., weights_only=True)`` only: a checkpoint whose ``recorder`` entry
holds objects that loader refuses is reported as such, never unpickled.

``Evaluator`` restates the PSNR of src/evaluators/nerf.py:22-25, 49-71 (``-10 log10(mean((pred - gt)^2))`` on the
(H, W, 3) image); SSIM / image files need skimage / cv2, which this image lacks, and are left out.
"""
from __future__ import annotations

import math
import os

import torch


def resolve_checkpoint(model_dir: str, epoch: int = -1) -> str | None:
    """The file ``load_network`` would open (net_utils.py:356-373), or None when there is none."""
    if not os.path.exists(model_dir):
        return None
    if not os.path.isdir(model_dir):
        return model_dir
    names = os.listdir(model_dir)
    pths = [int(n.split(".")[0]) for n in names if n.endswith(".pth") and n != "latest.pth" and n.split(".")[0].isdigit()]
    if not pths and "latest.pth" not in names:
        return None
    if epoch == -1:
        name = "latest" if "latest.pth" in names else str(max(pths))
    else:
        name = str(epoch)
    return os.path.join(model_dir, f"{name}.pth")


def load_network(net: torch.nn.Module, model_dir: str, epoch: int = -1, strict: bool = True, map_location="cpu") -> int:
    """``load_network`` (net_utils.py:348-381): loads ``['net']`` into ``net``; returns the epoch to resume at."""
    path = resolve_checkpoint(model_dir, epoch)
    if path is None:
        return 0
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=True)
    except Exception as e:  # the safe loader refused something in the file (e.g. a pickled recorder object)
        raise RuntimeError(f"{path}: torch.load(weights_only=True) refused this checkpoint ({e}); "
                           "re-save it with only tensors / plain containers (it is not unpickled here)") from e
    sd = ckpt["net"] if isinstance(ckpt, dict) and "net" in ckpt else ckpt
    result = net.load_state_dict(sd, strict=strict)
    if hasattr(net, "_packed_key"):
        net._packed_key = None
    if not strict and (result.missing_keys or result.unexpected_keys):
        print(f"load_network: missing {result.missing_keys}, unexpected {result.unexpected_keys}")
    return int(ckpt.get("epoch", -1)) + 1 if isinstance(ckpt, dict) else 0


def save_model(net: torch.nn.Module, model_dir: str, epoch: int, optim_state=None, scheduler_state=None,
               recorder_state=None, last: bool = False) -> str:
    """``save_model`` (net_utils.py:325-346): same dict layout and file names, so the reference reads it back."""
    os.makedirs(model_dir, exist_ok=True)
    model = {"net": {k: v.detach().cpu() for k, v in net.state_dict().items()}, "optim": optim_state or {},
             "scheduler": scheduler_state or {}, "recorder": recorder_state or {}, "epoch": int(epoch)}
    path = os.path.join(model_dir, "latest.pth" if last else f"{epoch}.pth")
    torch.save(model, path)
    pths = [int(n.split(".")[0]) for n in os.listdir(model_dir) if n != "latest.pth" and n.split(".")[0].isdigit()]
    if len(pths) > 5:                                        # keep the five newest numbered checkpoints (:342-346)
        os.remove(os.path.join(model_dir, f"{min(pths)}.pth"))
    return path


def psnr(img_pred: torch.Tensor, img_gt: torch.Tensor) -> float:
    """``Evaluator.psnr_metric`` (evaluators/nerf.py:22-25)."""
    mse = torch.mean((img_pred.float() - img_gt.float()) ** 2).item()
    return -10.0 * math.log(mse) / math.log(10.0)


class Evaluator:
    """``evaluate(output, batch)`` / ``summarize()`` of src/evaluators/nerf.py:49-95 (and evaluators/kilonerf.py,
    which reads ``rgb_map``), PSNR only."""

    def __init__(self, rgb_key: str = "fine_rgb_map"):
        self.rgb_key = rgb_key
        self.psnr = []
        self.imgs = []

    def evaluate(self, output: dict, batch: dict):
        H, W = int(batch["H"]), int(batch["W"])
        key = self.rgb_key if self.rgb_key in output else "rgb_map"
        img_pred = output[key].detach().reshape(H, W, 3)
        img_gt = batch["gt_rgb"].reshape(H, W, 3).to(img_pred.device)
        self.psnr.append(psnr(img_pred, img_gt))
        self.imgs.append(img_pred.cpu())
        return {}

    def summarize(self):
        if not self.psnr:
            return {}
        summary = {"mean_psnr": float(sum(self.psnr) / len(self.psnr))}
        self.psnr.clear()
        return summary
