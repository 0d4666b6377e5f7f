#!/usr/bin/env python3
"""Bisect of a non-repeatable cfg2 forward: runs geometry + forward N times under load and compares every saved intermediate
(geometry outputs, KNN maps, camera stages, point features, per-site fusion tensors, LiDAR stage outputs, FPN, head) bit for bit
with run 0; prints, per differing run, the tensors that changed in forward order.
Usage (GPU box): python tools/repeat_bisect.py [--dtype f32] [--runs 40] [--sibling]"""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--runs", type=int, default=40)
    ap.add_argument("--sibling", action="store_true")
    ap.add_argument("--sync-geometry", action="store_true", help="device-synchronise between the geometry and the forward")
    args = ap.parse_args()
    sib = None
    if args.sibling:
        sib = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "600", "--warmup", "2", "--no-cpu-baseline", "--no-roofline",
                                "--no-other-leg", "--no-batch-sweep", "--input", "resident"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    import torch
    from _util import pkg
    from test_gpu_benchsize import _cfg2_config
    det, calib, D, T = pkg("detfill"), pkg("calib"), pkg("data_import_carla"), pkg("train")
    lim6 = (0.0, 70.4, -40.0, 40.0, -2.4, 0.8)
    pts = [torch.from_numpy(det.synthetic_points(100000, lim6, 41 + b)).cuda() for b in range(2)]
    img = torch.stack([torch.from_numpy(det.synthetic_image(375, 1242, 41 + b)) for b in range(2)], 0).cuda()
    cfg = _cfg2_config(args.dtype, batch=2)
    cfg["hip_graphs"] = False
    tr = T.Train(cfg)
    det.fill_state_dict(tr.model)
    geo = D.FrameGeometry(cfg, calib.kitti_like_crt())
    hs = torch.cuda.Stream()
    ha = torch.empty(128 << 20, dtype=torch.float32, device="cuda"); hb = torch.empty_like(ha)

    def snapshot(x_lidar, geom, pred):
        ctx = tr.model._plan.ctx
        out = [("x_lidar", x_lidar), ("xyz", geom["xyz"]), ("uv", geom["uv"]), ("cnt", geom["cnt"])]
        out += [("knn%d" % i, t) for i, t in enumerate(geom["idx"])]
        im = ctx.get("img")
        if im:
            out += [("img.c1", im["c1"])] + [("img.c%d" % (i + 2), f) for i, f in enumerate(im["feats"])] + [("img.p2", im["p2"])]
        if "fmap" in cap:
            out.append(("fmap", cap["fmap"]))
        if "fuse_fp" in ctx:
            out.append(("fuse_fp", ctx["fuse_fp"]))
        for s in range(4):
            f = ctx.get("fuse%d" % s)
            if f:
                out += [("site%d.P" % s, f["P"]), ("site%d.hsum" % s, f["hsum"]), ("site%d.cnt" % s, f["cnt"])]
        for k in ("x2", "x3", "x4", "t2", "xp", "head"):
            if k in ctx:
                out.append((k, ctx[k]))
        out.append(("pred", pred))
        return [(n, t.detach().clone()) for n, t in out if torch.is_tensor(t)]

    plan = tr.model._plan
    orig_if = plan._image_forward
    cap = {}

    def image_forward(K, x_image, save):
        f = orig_if(K, x_image, save)
        cap["fmap"] = f
        return f
    plan._image_forward = image_forward
    ref = None
    nbad = 0
    for run in range(args.runs + 1):
        if run > 0 and not args.sibling:
            with torch.cuda.stream(hs):
                for _ in range(6):
                    hb.copy_(ha, non_blocking=True); ha.copy_(hb, non_blocking=True)
        x_lidar, geom = tr.geometry_async(geo, pts)
        if args.sync_geometry:
            torch.cuda.synchronize()
        pred = tr.model(x_lidar, img, geom=geom)
        torch.cuda.synchronize()
        snap = snapshot(x_lidar, geom, pred)
        if ref is None:
            ref = snap
            print("tensors compared:", " ".join(n for n, _ in ref))
            continue
        diff = [n for (n, a), (_, b) in zip(snap, ref) if a.shape != b.shape or not torch.equal(a, b)]
        if diff:
            nbad += 1
            print("run %d differs in: %s" % (run, " ".join(diff)), flush=True)
            d0 = dict(snap); r0 = dict(ref)
            if "fuse_fp" in diff and "fmap" in d0:
                # which tensor do the differing rows look like they were sampled from?
                import importlib
                ops = importlib.import_module("deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd.ops")
                fm, uvt, cn = d0["fmap"], d0["uv"], d0["cnt"]
                nmax = d0["fuse_fp"].shape[1]
                for f in range(fm.shape[0]):
                    rows = (d0["fuse_fp"][f] != r0["fuse_fp"][f]).any(dim=1).nonzero().flatten()
                    if not rows.numel():
                        continue
                    cands = {"fmap[this frame] now": ops.point_sample_fwd(tr.model.dtype, fm[f], uvt[f], cn[f:f + 1], nmax),
                             "fmap[other frame]": ops.point_sample_fwd(tr.model.dtype, fm[1 - f], uvt[f], cn[f:f + 1], nmax),
                             "p2[this frame] (unsmoothed)": ops.point_sample_fwd(tr.model.dtype, d0["img.p2"][f].contiguous(), uvt[f], cn[f:f + 1], nmax)}
                    torch.cuda.synchronize()
                    got = d0["fuse_fp"][f][rows]
                    print("    frame %d, %d differing rows: %s" % (f, rows.numel(), "; ".join("%s: %d rows equal" % (k, int((v[rows] == got).all(dim=1).sum())) for k, v in cands.items())))
                    r = int(rows[0])
                    print("    row %d run: %s" % (r, [round(v, 4) for v in d0["fuse_fp"][f][r][:8].float().cpu().tolist()]))
                    print("    row %d ref: %s" % (r, [round(v, 4) for v in r0["fuse_fp"][f][r][:8].float().cpu().tolist()]))
                    print("    row %d now: %s" % (r, [round(v, 4) for v in cands["fmap[this frame] now"][r][:8].float().cpu().tolist()]))
                    nd = (d0["fuse_fp"][f][rows] != r0["fuse_fp"][f][rows]).sum(dim=1)
                    print("    differing channels per differing row (of %d): %s" % (d0["fuse_fp"].shape[2], nd[:12].cpu().tolist()))
                    uvr = uvt[f][rows[:6]].cpu().tolist()
                    print("    their (u, v):", [(round(a, 1), round(b, 1)) for a, b in uvr])
            for n in diff[:2]:
                a, b = d0[n].float(), r0[n].float()
                ne = (a != b)
                if a.dim() == 3:                     # [B, rows, C]
                    for f in range(a.shape[0]):
                        rows = ne[f].any(dim=1).nonzero().flatten()
                        if rows.numel():
                            print("    %s frame %d: %d rows differ (first %d, last %d of %d; valid rows %d), max |diff| %.3g, run has zeros there: %s, ref has zeros there: %s" % (
                                n, f, rows.numel(), int(rows[0]), int(rows[-1]), a.shape[1], int(d0["cnt"][f]), float((a[f] - b[f]).abs().max()),
                                bool((a[f][rows] == 0).all()), bool((b[f][rows] == 0).all())), flush=True)
    print("%d of %d runs differ" % (nbad, args.runs))
    if sib is not None:
        sib.terminate()
        sib.wait(timeout=60)


if __name__ == "__main__":
    main()
