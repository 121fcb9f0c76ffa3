"""Ad-hoc GPU parity probe (development aid; the real tests live in tests/)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from robust_cvd_amd import api, synth
from robust_cvd_amd.ctypes_types import *
from oracle.oracle import Oracle

def compare(tag, video, ddesc, sdesc, intr=2, loss=1, solve=True, randomize=True, seed=0):
    p = OptParams.defaults(); p.num_threads = 8; p.intr_opt = intr; p.static_loss_type = loss
    rng = np.random.default_rng(seed)
    objs = {}
    for name, ctor in (("hip", lambda: api.Solver(0)), ("oracle", Oracle)):
        s = ctor(); synth.load_into(s, video)
        s.reset_depth_xforms(ddesc); s.reset_spatial_xforms(sdesc)
        objs[name] = s
    F = video.num_frames; B = objs["hip"].block_size()
    assert B == objs["oracle"].block_size()
    pose = objs["oracle"].get_pose_params()
    dx = objs["oracle"].get_xform_params(False); sx = objs["oracle"].get_xform_params(True)
    if randomize:
        pose[:, :3] = rng.normal(0, 0.05, (F, 3)); pose[:, 3:6] = rng.normal(0, 0.05, (F, 3)); pose[0, 3:6] = 0
        pose[:, 6] = 0.2 + rng.uniform(0, 0.05, F)
        if dx.size:
            dx = 0.15 + rng.uniform(0, 0.05, dx.shape)
            if ddesc.value_xform == 2: dx[:, 1::2] = rng.uniform(0, 0.3, dx[:, 1::2].shape)
        sx = rng.normal(0, 0.01, sx.shape)
    res = {}
    for name, s in objs.items():
        s.set_xform_params(dx, False); s.set_xform_params(sx, True)
        t0 = time.time()
        res[name] = s.evaluate(p, 0.1, pose, want_gradient=True, want_hdiag=True, want_hfull=(F * B <= 400))
        res[name]["t"] = time.time() - t0
    h, o = res["hip"], res["oracle"]
    gs = np.abs(o["gradient"]).max()
    line = (f"{tag}: B={B} nres {h['num_residual_blocks']}/{o['num_residual_blocks']} cost {h['cost']:.12g}/{o['cost']:.12g} "
            f"rel {abs(h['cost']-o['cost'])/abs(o['cost']):.2e} grad {np.abs(h['gradient']-o['gradient']).max()/gs:.2e} "
            f"hdiag {np.abs(h['hdiag']-o['hdiag']).max()/np.abs(o['hdiag']).max():.2e}")
    if h["hfull"] is not None:
        line += f" hfull {np.abs(h['hfull']-o['hfull']).max()/np.abs(o['hfull']).max():.2e}"
    print(line, flush=True)
    return objs, p

if __name__ == "__main__":
    v = synth.make_video(6, 64, 40, seed=3, spacing=9)
    compare("global/perframe", v, XformDesc.global_depth(), XformDesc.spatial())
    compare("global/fixed", v, XformDesc.global_depth(), XformDesc.spatial(), intr=0)
    compare("grid lin 4x3", v, XformDesc.grid_depth(4, 3), XformDesc.spatial())
    compare("grid cubic 5x4 ss + bicubic spatial", v, XformDesc.grid_depth(5, 4, ValueXformType.ScaleShift, cubic=True), XformDesc.spatial(SpatialXformType.BicubicGrid, 4, 3))
    compare("global ss euclid cornersbilinear", v, XformDesc.global_depth(ValueXformType.ScaleShift), XformDesc.spatial(SpatialXformType.CornersBilinear), loss=0)
    compare("grid lin ratio vertical", v, XformDesc.grid_depth(3, 3), XformDesc.spatial(SpatialXformType.VerticalLinear), loss=2)
    compare("grid lin log bilinear-spatial", v, XformDesc.grid_depth(3, 3), XformDesc.spatial(SpatialXformType.BilinearGrid, 3, 2), loss=3)
    # full solve comparison
    v2 = synth.make_video(12, 96, 56, seed=1)
    p = OptParams.defaults(); p.num_threads = 8
    out = {}
    for name, ctor in (("hip", lambda: api.Solver(0)), ("oracle", Oracle)):
        s = ctor(); synth.load_into(s, v2)
        s.reset_depth_xforms(XformDesc.global_depth()); s.reset_spatial_xforms(XformDesc.spatial())
        t0 = time.time(); s.normalize_depth(p); t1 = time.time()
        nd = s.get_xform_params()[:2].ravel()
        s.pose_optimization(p); t2 = time.time()
        out[name] = (s.get_poses(), s.get_xform_params(), s.summary(), s.xform_desc())
        print(name, "normalize", nd, f"{t1-t0:.3f}s", "poseopt", f"{t2-t1:.3f}s", s.summary(), flush=True)
    pe = synth.relative_pose_error(out["hip"][0]["position"], out["hip"][0]["orientation"], out["oracle"][0]["position"], out["oracle"][0]["orientation"])
    print("pose err", pe, "xform max rel diff", np.abs(out["hip"][1]-out["oracle"][1]).max()/np.abs(out["oracle"][1]).max(),
          "fov diff", np.abs(out["hip"][0]["vfov"]-out["oracle"][0]["vfov"]).max())
