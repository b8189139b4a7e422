"""Reader side of the distillation data format (DistDataset, kilonerf/blender.py:168-220) on CPU."""
import pytest
import torch

from nerf_amd.distill import DistRecords


def _write(tmp_path, n_views=3, n=50):
    for v in range(n_views):
        base = torch.arange(n, dtype=torch.float32)[:, None] + 1000 * v
        torch.save({"xyz": base.repeat(1, 3), "viewdirs": -base.repeat(1, 3), "teacher_rgb": base.repeat(1, 3) / 1e4,
                    "teacher_alpha": base / 1e4}, str(tmp_path / f"dist_{v:04d}.pt"))


def test_records_items_and_batches(tmp_path):
    _write(tmp_path)
    ds = DistRecords(str(tmp_path), n_rays=16, generator=torch.Generator().manual_seed(0))
    assert len(ds) == 3 and [f[-12:] for f in ds.files] == ["dist_0000.pt", "dist_0001.pt", "dist_0002.pt"]
    it = ds[1]
    assert set(it) == {"xyz", "viewdirs", "teacher_rgb", "teacher_alpha"}
    assert it["xyz"].shape == (16, 3) and it["teacher_alpha"].shape == (16, 1)
    ids = it["xyz"][:, 0] - 1000
    assert len(set(ids.tolist())) == 16 and ids.min() >= 0 and ids.max() < 50        # without replacement, view 1
    assert torch.equal(it["viewdirs"], -it["xyz"]) and torch.equal(it["teacher_alpha"][:, 0], it["xyz"][:, 0] / 1e4)
    b = ds.batch([2, 0, 1, 2])                                                       # batch_size 4 (lego.yaml:120)
    assert b["xyz"].shape == (4, 16, 3) and b["teacher_alpha"].shape == (4, 16, 1)
    assert (b["xyz"][0, :, 0] >= 2000).all() and (b["xyz"][1, :, 0] < 1000).all()


def test_records_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        DistRecords(str(tmp_path))
    _write(tmp_path, 1, 8)
    with pytest.raises(ValueError):
        DistRecords(str(tmp_path), n_rays=16)[0]
    torch.save({"xyz": torch.zeros(4, 3)}, str(tmp_path / "dist_0009.pt"))
    with pytest.raises(KeyError):
        DistRecords(str(tmp_path))
