"""Force-map parity: CPU oracle == reference goldens bit-for-bit (sha256 of the full 194 MB tensor);
HIP kernel vs the same goldens on the GPU (<= 1 bf16 ulp on a vanishing fraction: device expf)."""
import hashlib
import os

import numpy as np
import pytest
import torch

import gen_inputs as gi
from conftest import GOLDEN
from oracle import force_map_oracle as fo

META = ["force", "angle", "x_pos", "y_pos", "tforce", "tangle", "tx", "ty", "pmass", "tmass", "pcx", "pcy", "tcx", "tcy"]


def _rows():
    g = np.load(os.path.join(GOLDEN, "g7_force_maps.npz"))
    return g, [dict(zip(META, r)) for r in g["meta"]]


def _oracle(r):
    return fo.control_video(r["force"], r["angle"], r["x_pos"], r["y_pos"], r["tforce"], r["tangle"], r["tx"], r["ty"],
                            {"projectile": r["pmass"], "target": r["tmass"], "distractors": []},
                            {"projectile": [int(r["pcx"]), int(r["pcy"])], "target": [int(r["tcx"]), int(r["tcy"])],
                             "distractors": []})


@pytest.mark.parametrize("k", [0, 8, 12, 13])
def test_oracle_matches_reference_sha256(k):
    g, rows = _rows()
    cv = _oracle(rows[k])
    raw = gi.to_u16(cv)
    assert hashlib.sha256(raw.tobytes()).digest() == bytes(g[f"sha256_{k}"]), g["names"][k]
    assert np.array_equal(raw[::8, ::8, ::8, :], g[f"down_{k}"])


def test_plan_matches_oracle_geometry():
    """host BlobPlan (float64 scalar math) places the same blobs: centre pixel of the goal-force blob in
    the oracle's frame 0 / 80 is its argmax."""
    from goal_force_amd.force_map import plan_control_video
    g, rows = _rows()
    r = rows[13]
    np.random.seed(0)
    plan = plan_control_video(r["force"], r["angle"], r["x_pos"], r["y_pos"], r["tforce"], r["tangle"], r["tx"], r["ty"],
                              81, 480, 832, {"projectile": r["pmass"], "target": r["tmass"], "distractors": []},
                              {"projectile": [int(r["pcx"]), int(r["pcy"])], "target": [int(r["tcx"]), int(r["tcy"])],
                               "distractors": []}, 30.0, 400.0, 30.0, 400.0, 1.0, 4.0)
    ch, pr, ce = plan.arrays()
    assert list(ch) == [0, 1, 2, 2] and plan.clamp01
    assert np.allclose(pr[:2, 0], 800.0) and pr.shape == (4, 2) and ce.shape == (4, 81, 2)
    cv = _oracle(r).float()
    for b in (0, 1):
        for fr in (0, 40, 80):
            idx = int(cv[fr, :, :, b].argmax())
            yy, xx = divmod(idx, 832)
            cx, cy = ce[b, fr]
            if 0 <= cx < 832 and 0 <= cy < 480:
                assert abs(xx - cx) <= 3.0 and abs(yy - cy) <= 3.0  # bf16 plateau around the peak


@pytest.mark.gpu
@pytest.mark.parametrize("k", [0, 3, 8, 12, 13])
def test_hip_force_map_vs_reference_golden(k):
    from goal_force_amd.force_map import plan_control_video, render_control_video
    g, rows = _rows()
    r = rows[k]
    np.random.seed(0)
    plan = plan_control_video(r["force"], r["angle"], r["x_pos"], r["y_pos"], r["tforce"], r["tangle"], r["tx"], r["ty"],
                              81, 480, 832, {"projectile": r["pmass"], "target": r["tmass"], "distractors": []},
                              {"projectile": [int(r["pcx"]), int(r["pcy"])], "target": [int(r["tcx"]), int(r["tcy"])],
                               "distractors": []}, 30.0, 400.0, 30.0, 400.0, 1.0, 4.0)
    got = render_control_video(plan).cpu()
    assert got.shape == (81, 480, 832, 3) and got.dtype == torch.bfloat16
    raw = gi.to_u16(got)
    for name, sl in (("down", raw[::8, ::8, ::8, :]), ("frame40", raw[40, 150:330:2, ::4, :])):
        ref = g[f"{name}_{k}"]
        d = np.abs(sl.astype(np.int32) - ref.astype(np.int32))
        assert d.max() <= 1, f"{name}: max ulp diff {d.max()}"
        assert (d > 0).mean() < 2e-3, f"{name}: {(d > 0).mean():.2e} of elements differ by 1 ulp"
    cs = got.float().sum(dim=(0, 1, 2)).numpy()
    assert np.allclose(cs, g[f"chan_sum_{k}"], rtol=2e-4, atol=1.0)
    if hashlib.sha256(raw.tobytes()).digest() == bytes(g[f"sha256_{k}"]):
        print(f"row {k}: bit-exact with the reference")


# ------------------------------------------------------------------ the dominos / plants dataset variants (DS:1099-1894; g15)
def _g15():
    return np.load(os.path.join(GOLDEN, "g15_dataset_variants.npz"))


def test_dominos_golden_is_the_balls_golden():
    """The reference's ControlSignalDataset_Dominos._generate_control_video (DS:1253-1367) is a copy of the balls one: run on g7's
    synthetic rows by make_goldens.py::g15 it produced g7's bytes — which is why the product's Dominos class is the Balls renderer."""
    from goal_force_amd.force_map import ControlSignalDataset_Balls, ControlSignalDataset_Dominos
    g, g7 = _g15(), np.load(os.path.join(GOLDEN, "g7_force_maps.npz"))
    for k in (12, 13):
        assert bytes(g[f"dominos_sha256_{k}"]) == bytes(g7[f"sha256_{k}"])
    assert issubclass(ControlSignalDataset_Dominos, ControlSignalDataset_Balls)
    assert ControlSignalDataset_Dominos._generate_control_video is ControlSignalDataset_Balls._generate_control_video


@pytest.mark.parametrize("k", range(len(gi.PLANTS_ROWS)))
def test_plants_oracle_matches_reference_sha256(k):
    g = _g15()
    f, a, x, y, nf, h, w = gi.PLANTS_ROWS[k]
    raw = gi.to_u16(fo.plants_control_video(f, a, x, y, num_frames=nf, height=h, width=w))
    assert hashlib.sha256(raw.tobytes()).digest() == bytes(g[f"plants_sha256_{k}"])
    assert np.array_equal(raw[::8, ::8, ::8, :], g[f"plants_down_{k}"])


def test_plants_plan_is_one_unclamped_direct_force_blob():
    from goal_force_amd._lib import GoalForceError
    from goal_force_amd.force_map import ControlSignalDataset_Plants
    ds = object.__new__(ControlSignalDataset_Plants)
    ds.min_force, ds.max_force = 30.0, 400.0
    f, a, x, y, nf, h, w = gi.PLANTS_ROWS[3]
    plan = ds.plan(f, a, x, y, nf, h, w)
    ch, pr, ce = plan.arrays()
    assert list(ch) == [0] and not plan.clamp01 and pr.shape == (1, 2) and float(pr[0, 0]) == 800.0 and ce.shape == (1, 49, 2)
    disp = w / 8 + (w / 2 - w / 8) * ((f - 30.0) / 370.0)
    assert np.allclose(ce[0, 0], [x * w, (1 - y) * h]) and np.allclose(ce[0, -1, 0], x * w + disp * np.cos(np.deg2rad(a)), atol=1e-3)
    with pytest.raises(GoalForceError):
        ds.plan(-1, a, x, y, nf, h, w)


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(gi.PLANTS_ROWS)))
def test_hip_plants_force_map_vs_reference_golden(k):
    from goal_force_amd.force_map import ControlSignalDataset_Plants
    g = _g15()
    ds = object.__new__(ControlSignalDataset_Plants)
    ds.min_force, ds.max_force, ds.device = 30.0, 400.0, "cuda"
    f, a, x, y, nf, h, w = gi.PLANTS_ROWS[k]
    got = ds._generate_control_video(f, a, x, y, num_frames=nf, num_channels=3, height=h, width=w).cpu()
    assert got.shape == (nf, h, w, 3) and got.dtype == torch.bfloat16 and float(got[..., 1:].abs().max()) == 0
    raw = gi.to_u16(got)
    for name, sl in (("down", raw[::8, ::8, ::8, :]), ("frame", raw[nf // 2, ::2, ::4, :])):
        d = np.abs(sl.astype(np.int32) - g[f"plants_{name}_{k}"].astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, f"{name}: max ulp diff {d.max()}, {(d > 0).mean():.2e} differ"
    assert np.allclose(got.float().sum(dim=(0, 1, 2)).numpy(), g[f"plants_chan_sum_{k}"], rtol=2e-4, atol=1.0)


@pytest.mark.gpu
def test_hip_plants_and_dominos_datasets_read_their_csv_rows(tmp_path):
    """Validation-mode datasets end to end: CSV row + PNG -> the dict the drivers consume (DS:1488-1539, 1842-1878)."""
    from PIL import Image
    from goal_force_amd.force_map import ControlSignalDataset_Dominos, ControlSignalDataset_Plants
    (tmp_path / "images").mkdir()
    Image.fromarray(np.zeros((480, 832, 3), np.uint8)).save(tmp_path / "images" / "fern.png")
    (tmp_path / "plants.csv").write_text('image,force,angle,coordx,coordy,width,height,caption\nfern.png,200.0,45.0,208,120,832,480,"The fern sways."\n'
                                         'missing.png,1,1,1,1,832,480,"dropped: no such image"\n')
    ds = ControlSignalDataset_Plants(base_path=str(tmp_path), metadata_path=str(tmp_path / "plants.csv"), is_validation_dataset=True,
                                     num_frames=81, height=480, width=832)
    assert len(ds) == 1 and (ds.min_force, ds.max_force) == (0.0, 1.0)
    ds.min_force, ds.max_force = 30.0, 400.0
    d = ds[0]
    assert d["prompt"] == "The fern sways." and d["file_id"] == "fern" and len(d["video"]) == 1 and d["x_pos"] == 0.25 and d["y_pos"] == 0.25
    g = _g15()          # PLANTS_ROWS[0] is this row
    raw = gi.to_u16(d["control_video"].cpu())
    dd = np.abs(raw[::8, ::8, ::8, :].astype(np.int32) - g["plants_down_0"].astype(np.int32))
    assert dd.max() <= 1
    (tmp_path / "dom.csv").write_text("image,projectile_force_angle,projectile_force_magnitude,projectile_coordx,projectile_coordy,projectile_mass,"
                                      "target_indirect_force_angle,target_indirect_force_magnitude,target_coordx,target_coordy,target_mass,width,height,caption\n"
                                      'fern.png,45.0,200.0,208,120,2.0,-1.0,-1.0,600,300,-1.0,832,480,"The domino falls."\n')
    dm = ControlSignalDataset_Dominos(base_path=str(tmp_path), metadata_path=str(tmp_path / "dom.csv"), is_validation_dataset=True,
                                      num_frames=81, height=480, width=832)
    dm.min_mass, dm.max_mass, dm.min_force, dm.max_force = 1.0, 4.0, 30.0, 400.0
    dm.min_indirect_force, dm.max_indirect_force = 30.0, 400.0
    e = dm[0]
    g7 = np.load(os.path.join(GOLDEN, "g7_force_maps.npz"))          # its synthetic row 12 is this row
    raw = gi.to_u16(e["control_video"].cpu())
    dd = np.abs(raw[::8, ::8, ::8, :].astype(np.int32) - g7["down_12"].astype(np.int32))
    assert dd.max() <= 1 and e["prompt"] == "The domino falls."


def _row13():
    g7 = np.load(os.path.join(GOLDEN, "g7_force_maps.npz"))
    return dict(zip(META, g7["meta"][13]))


@pytest.mark.parametrize("seed", range(8))
def test_training_time_channel_masking_follows_the_reference_draws(seed):
    """DS:785-802, 851-853 with the masking probabilities on (0.25 / 0.25 / 0.5): which of the three channels survive is decided by
    np.random draws — g15 ran the reference's class under np.random.seed(seed); the oracle (sha256) and the product's plan (the same
    channels alive, the clamp only when the masses are drawn) consume the same draws in the same order."""
    from goal_force_amd.force_map import plan_control_video
    g, r = _g15(), _row13()
    masses = {"projectile": r["pmass"], "target": r["tmass"], "distractors": []}
    coords = {"projectile": [int(r["pcx"]), int(r["pcy"])], "target": [int(r["tcx"]), int(r["tcy"])], "distractors": []}
    np.random.seed(seed)
    cv = fo.control_video(r["force"], r["angle"], r["x_pos"], r["y_pos"], r["tforce"], r["tangle"], r["tx"], r["ty"], masses, coords, num_frames=9,
                          p_mask_out_direct_force=0.25, p_mask_out_indirect_force=0.25, p_mask_out_masses=0.5)
    assert hashlib.sha256(gi.to_u16(cv).tobytes()).digest() == bytes(g[f"masked_sha256_{seed}"])
    np.random.seed(seed)
    plan = plan_control_video(r["force"], r["angle"], r["x_pos"], r["y_pos"], r["tforce"], r["tangle"], r["tx"], r["ty"], 9, 480, 832, masses, coords,
                              30.0, 400.0, 30.0, 400.0, 1.0, 4.0, 0.25, 0.25, 0.5)
    alive = [int(c in plan.channels) for c in range(3)]
    assert alive == (g["masked_chan_sums"][seed] > 0).astype(int).tolist() and plan.clamp01 == bool(alive[2])


@pytest.mark.gpu
def test_hip_masked_control_videos_vs_reference_golden():
    from goal_force_amd.force_map import plan_control_video, render_control_video
    g, r = _g15(), _row13()
    masses = {"projectile": r["pmass"], "target": r["tmass"], "distractors": []}
    coords = {"projectile": [int(r["pcx"]), int(r["pcy"])], "target": [int(r["tcx"]), int(r["tcy"])], "distractors": []}
    for seed in range(8):
        np.random.seed(seed)
        plan = plan_control_video(r["force"], r["angle"], r["x_pos"], r["y_pos"], r["tforce"], r["tangle"], r["tx"], r["ty"], 9, 480, 832, masses,
                                  coords, 30.0, 400.0, 30.0, 400.0, 1.0, 4.0, 0.25, 0.25, 0.5)
        raw = gi.to_u16(render_control_video(plan).cpu())
        d = np.abs(raw[::4, ::8, ::8, :].astype(np.int32) - g[f"masked_down_{seed}"].astype(np.int32))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, (seed, int(d.max()))


@pytest.mark.gpu
def test_hip_training_mode_items_of_the_dataset_mux(tmp_path):
    """`training.get_dataset(args)` in training mode end to end (train.py:126-197 -> DS:1029-1080, 1488-1539, 1842-1878): every item a
    list of num_frames PIL frames + the control video; the carnation clip goes through the random zoom-crop (DS:1773-1834) with the
    poke point re-expressed in the crop."""
    from goal_force_amd import training as tr
    gi.write_training_tree(str(tmp_path))
    rng = np.random.default_rng(3)
    for folder, names, n in (("balls", ["b0", "b1", "b2"], 24), ("dominos", ["d0", "d1"], 30), ("plants", ["fern0", "carnation1"], 12)):
        for nm in names:
            os.remove(tmp_path / folder / f"{nm}.mp4")
            np.save(tmp_path / folder / f"{nm}.npy", rng.integers(0, 255, (n, 480, 832, 3), dtype=np.uint8))
            os.rename(tmp_path / folder / f"{nm}.npy", tmp_path / folder / f"{nm}.mp4")          # the listing names .mp4 files

    def loader(path):                        # an .npy behind the listing's name (no video decoder in this image)
        from PIL import Image
        return [Image.fromarray(a) for a in np.load(open(path, "rb"))]
    argv = gi.training_cli(str(tmp_path))
    argv[argv.index("--num_frames") + 1] = "9"
    args = tr.wan_parser().parse_args(argv)
    ds = tr.get_dataset(args, device="cuda", video_loader=loader)
    assert len(ds) == 7
    np.random.seed(0)
    items = [ds[i] for i in range(7)]
    for it in items:
        assert len(it["video"]) == 9 and it["video"][0].size == (832, 480) and tuple(it["control_video"].shape) == (9, 480, 832, 3)
        assert it["control_video"].dtype == torch.bfloat16 and it["control_video"].is_cuda
        assert tr.data_is_correct_shape_and_type(it, "direct_force_and_goal_force_and_mass", 9)
    raw = np.load(open(tmp_path / "balls" / "b0.mp4", "rb"))
    want = raw[::2][-9:][0].astype(int)
    want[(want >= 1) & (want <= 63)] -= 1
    assert np.array_equal(np.array(items[0]["video"][0]).astype(int), want), "balls: every second frame, the last 9, through the pixel round trip"
    raw = np.load(open(tmp_path / "dominos" / "d0.mp4", "rb"))
    want = raw[14:][0:9][8].astype(int)
    want[(want >= 1) & (want <= 63)] -= 1
    assert np.array_equal(np.array(items[3]["video"][8]).astype(int), want)
    fern, carn = items[5], items[6]
    assert fern["file_id"] == "fern0" and fern["x_pos"] == 100 / 832 and carn["file_id"] == "carnation1"
    assert float(fern["control_video"][..., 1:].abs().max()) == 0 and float(carn["control_video"][..., 0].max()) > 0.9
    assert 50 / 832 <= carn["x_pos"] <= 1 - 50 / 832 + 0.1 and 0 < carn["y_pos"] < 1, "the poke point stays inside the zoomed window"
