#!/usr/bin/env python3
"""Benchmark of the continuous-fusion train step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (no launcher: starts its N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One step = one pass of the whole hot path over one batch of synthetic frames.  By default (--input host, round 5) the frames
start in HOST memory and reach HBM through frame_loader.FrameLoader's staging thread and copy stream, up to four batches ahead of
the step that consumes them (the metric starts at the H2D copy of the raw frame), so every step after the first
finds its inputs resident when it starts; --input resident keeps the raw clouds and images in HBM (rounds 1-4's headline; it runs
as a short second leg and is reported beside `value` either way).  The path: voxelise + project (raw 100k-point clouds), KNN for the 4 fusion sites, ResNet-18 image
stream + FPN, LiDAR-BEV stream with 4 continuous-fusion adds, heads/decode, LossTotal,
backward, gradient all-reduce (RCCL, N>1) and the fused Adam step.  Workload = BASELINE.json
configs[1] ("cfg2"): KITTI-scale grid 32x704x800, 100k points, 1242x375 RGB, ResNet-18, K=3,
batch 2 per GPU, bf16 (fp32 accumulate / fp32 master weights).  Weak scaling: per-GPU batch fixed.

Prints ONE JSON line on rank 0 (see README / task contract), including
  roofline     -- the dominant kernel class, timed live with HIP events on the launch stream
  cpu_baseline -- this repo's CPU restatement (oracle/, kind "port") on a bounded sample, rank 0, N=1 only.
"""
import argparse
import copy
import importlib
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd"


def pkg(sub):
    return importlib.import_module(PKG + "." + sub)


def kitti_config(batch, dtype="bf16", n_points=100000, K=3, image_stream="resnet18", image_wh=(1242, 375)):
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, PKG, "config", "config_carla.yaml")))
    cfg.update(dict(voxel_length=704, voxel_width=800, voxel_channel=32, lidar_x_min=0.0, lidar_x_max=70.4,
                    lidar_y_min=-40.0, lidar_y_max=40.0, lidar_z_min=-2.4, lidar_z_max=0.8,
                    image_height=image_wh[1], image_width=image_wh[0], max_num_pc=n_points, batch_size=batch,
                    dtype=dtype, projection_mode="correct", voxel_mode="compat"))
    cfg["fusion"] = dict(enabled=True, K=K, r_max=None, image_channels=64, image_stream=image_stream, zero_init_last=False)
    return cfg


class FramePool(object):
    """Synthetic frames (SURVEY.md 8(d) generator), uploaded once; the timed region starts from HBM."""

    def __init__(self, cfg, n_frames, n_points, seed0):
        D = pkg("data_import_carla")
        calib = pkg("calib")
        crt = calib.hd_crt() if cfg["image_width"] == 1920 else calib.kitti_like_crt()     # SURVEY.md 8(d) calibrations
        ds = D.SyntheticDataset(cfg, length=n_frames, num_points=n_points, crt=crt,
                                image_hw=(cfg["image_height"], cfg["image_width"]))
        self.geometry = ds.geometry
        self.pts, self.img, self.boxes, self.nb = [], [], [], []
        for i in range(n_frames):
            p, im, b, nb = ds.raw_frame(seed0 + i)
            self.pts.append(p.cuda())
            self.img.append(im.cuda())
            self.boxes.append(b)
            self.nb.append(nb)
        self.boxes_dev = [b.cuda() for b in self.boxes]
        self.n = n_frames
        self._img_batches = {}

    def batch(self, step, B):
        ids = [(step * B + i) % self.n for i in range(B)]
        return ids

    def image_batch(self, ids):
        """[B,3,H,W] uint8 batch tensor, resident in HBM like the clouds (a loader collates frames into one tensor: FrameLoader
        does it in pinned host memory; here the few distinct batches of the pool are stacked once)."""
        key = tuple(ids)
        if key not in self._img_batches:
            self._img_batches[key] = torch.stack([self.img[i] for i in ids], 0)
        return self._img_batches[key]


class HostFrames(torch.utils.data.Dataset):
    """The pool's frames as HOST tensors in the datasets' raw-mode contract (--from-host: the step then includes pinned
    staging + the PCIe copies of frame_loader.FrameLoader: the bench's default input since round 5)."""
    raw = True

    def __init__(self, pool, steps, B):
        self.pts = [p.cpu() for p in pool.pts]
        self.img = [i.cpu() for i in pool.img]
        self.boxes, self.nb, self.n, self.length = pool.boxes, pool.nb, pool.n, steps * B

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        k = i % self.n
        return {"image": self.img[k], "bboxes": self.boxes[k], "num_bboxes": self.nb[k], "lidar_points": self.pts[k], "crt": None}


def train_step(trainer, pool, ids):
    """Whole hot path for one batch.  Geometry + KNN run on the trainer's side stream (overlapping the camera
    stream on the compute stream); everything else is enqueued on torch's current stream."""
    x_lidar, geom = trainer.geometry_async(pool.geometry, [pool.pts[i] for i in ids])
    x_image = pool.image_batch(ids)
    boxes = torch.stack([pool.boxes[i] for i in ids], 0)          # CPU, as a DataLoader would hand them over
    nb = torch.tensor([pool.nb[i] for i in ids])
    trainer.one_step(x_lidar, x_image, boxes, nb, geom=geom)


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s HBM3E (6.3 TB/s measured with a float4 copy)


def mfma_peak(name):
    """Dense MFMA peak of the element type a kernel name carries (TFLOP/s), MI355X_MICROARCH.md."""
    return 5000.0 if "fp8" in name else (2500.0 if ("bf16" in name or "f16" in name) else 157.3)


KERNEL_CLASSES = (      # (class, name prefixes) -- SURVEY.md 8(d) "binding roof per class"
    ("conv forward (MFMA implicit GEMM)", ("conv_fwd", "stem_fwd")),
    ("conv input gradient", ("conv_dgrad",)),
    ("conv weight gradient", ("conv_wgrad", "stem_wgrad")),
    ("weight prep / slab finalisation / Adam", ("weight_prep", "wgrad_finalize", "adam")),
    ("voxelise + project + compact", ("voxel_", "project_", "compact_", "range_")),
    ("KNN (sort + search + inverse maps)", ("knn_", "scan_", "inv_")),
    ("fusion gather / point sampling", ("fusion_", "point_sample", "rowscale_")),
    ("elementwise (resize, pool, ReLU mask, head, casts, loss)", ("resize_", "maxpool_", "relu_", "head_", "cast", "loss_", "nchw_", "nhwc_", "image_")),
)


def roofline_leg(trainer, pool, B, steps):
    """Instrumented pass of the same train step: libdcf_hip brackets every launch with HIP events on the
    launch stream and records the launch's ALGORITHMIC work as the launch itself declares it inside the library:
    flops (conv kernels: 2*M*Cout*Cin*taps; dgrad is priced at the forward conv's flops) and / or bytes (the tensors a
    kernel has to read and write once, whatever its tiling re-reads).  Kernel names are template
    instantiations, so the average durations line up with `rocprofv3 --kernel-trace --stats` rows (profiles/)."""
    Hm = pkg("_hip")
    trainer.model.graphs_off = True                    # per-launch event brackets need eager launches
    Hm.call("dcf_prof_reset")
    Hm.call("dcf_prof_enable", 1)
    for s in range(steps):
        train_step(trainer, pool, pool.batch(1000 + s, B))
    Hm.call("dcf_prof_calibrate", Hm.stream_ptr(), 200)
    torch.cuda.synchronize()
    Hm.call("dcf_prof_enable", 0)
    trainer.model.graphs_off = False
    prof = Hm.prof_read()
    Hm.call("dcf_prof_reset")
    empty = prof.pop("__empty_bracket__", (0.0, 1, 0.0, 0.0))
    bracket_ms = empty[0] / max(empty[1], 1)             # cost of the event pair itself, subtracted per launch
    # entry: [ms with the bracket cost taken out, launches, flops, bytes, ms as measured]
    prof = {n: [max(v[0] - bracket_ms * v[1], 1e-9), v[1], v[2], v[3], v[0]] for n, v in prof.items()}
    # table-driven launches cannot see their sizes inside the library: priced here from the model
    K = trainer.model._backend
    P = float(trainer.model.flat_params.numel())
    if "weight_prep" in prof:
        prof["weight_prep"][3] = prof["weight_prep"][1] * (P * 4.0 + float(K.warena.numel()))
    if "wgrad_finalize" in prof and K.slabs is not None:
        prof["wgrad_finalize"][3] = prof["wgrad_finalize"][1] * (float(K.slabs.numel()) * 4.0 + P * 8.0)
    total_ms = sum(v[0] for v in prof.values())

    def rates(name, ms, work, byts):
        tf = work / (ms * 1e-3) / 1e12 if work > 0 else None
        gb = byts / (ms * 1e-3) / 1e9 if byts > 0 else None
        fm = tf / mfma_peak(name) if tf is not None else None
        fh = gb / HBM_PEAK_GBS if gb is not None else None
        if fm is None and fh is None:
            return tf, gb, None, None
        bound = "mfma" if (fh is None or (fm is not None and fm >= fh)) else "hbm"
        return tf, gb, bound, (fm if bound == "mfma" else fh)

    table = sorted(((n, v[0], v[1], v[2], v[3], v[4]) for n, v in prof.items()), key=lambda t: -t[1])
    # The dominant kernel is the GPU FUNCTION with the most time, as rocprofv3's kernel stats count it: launch names that
    # differ only in run-time arguments (the row-sharing convolution's position tiles per workgroup; forward / input
    # gradient = the same function on two weight images) are one row there, so they are summed here too.
    # Ranked by the time AS MEASURED (bracket cost included): subtracting ~4.5 us per launch first would push functions made of
    # many short launches down the list and name another row than rocprofv3's kernel stats do.
    groups = {}
    for n, m, c, w, by, raw in table:
        rk = rocprof_kernel(n)
        key = "%s<%s>" % (rk[0], ",".join(rk[1])) if rk else n
        g = groups.setdefault(key, [n, 0.0, 0, 0.0, 0.0, [], 0.0])
        g[1] += m; g[2] += c; g[3] += w; g[4] += by; g[5].append(n); g[6] += raw
    symbol, (name, ms, calls, work, byts, members, raw_ms) = max(groups.items(), key=lambda kv: kv[1][6])
    tf, gb, bound, frac = rates(name, ms, work, byts)
    if bound == "hbm":
        roof = {"kernel": name, "bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(frac, 4), "traffic": None}
    else:
        roof = {"kernel": name, "bound": "mfma", "achieved": round(tf, 2) if tf else None, "peak": mfma_peak(name), "unit": "TFLOP/s",
                "frac": round(frac, 4) if frac else None, "traffic": None}
    roof.update({"avg_launch_us": round(ms * 1e3 / max(calls, 1), 2), "launches_per_step": calls / steps,
                 "share_of_gpu_time": round(ms / total_ms, 3) if total_ms else None, "flops_per_step": work / steps,
                 "algorithmic_bytes_per_launch": round(byts / max(calls, 1)) if byts else None,
                 "gpu_ms_per_step_all_kernels": round(total_ms / steps, 3), "event_bracket_us_subtracted": round(bracket_ms * 1e3, 2),
                 "avg_launch_us_with_bracket": round(raw_ms * 1e3 / max(calls, 1), 2), "ranked_by": "time as measured (bracket included)"})
    roof["traffic"], roof["traffic_source"], roof["traffic_stale"] = pmc_traffic(name)
    roof["kernel"] = symbol
    roof["launch_names"] = members
    breakdown = []
    for n, m, c, w, by, _raw in table[:48]:
        tf, gb, bound, frac = rates(n, m, w, by)
        breakdown.append({"kernel": n, "ms_per_step": round(m / steps, 4), "calls_per_step": c / steps, "tflops": round(tf, 1) if tf else None,
                          "gbps": round(gb, 1) if gb else None, "bound": bound, "frac": round(frac, 4) if frac else None})
    classes = []
    for cname, prefixes in KERNEL_CLASSES:
        rows = [t for t in table if t[0].startswith(prefixes)]
        if not rows:
            continue
        m = sum(t[1] for t in rows); w = sum(t[3] for t in rows); by = sum(t[4] for t in rows)
        pk = max(mfma_peak(t[0]) for t in rows)
        tf = w / (m * 1e-3) / 1e12 if w > 0 else None
        gb = by / (m * 1e-3) / 1e9 if by > 0 else None
        classes.append({"class": cname, "ms_per_step": round(m / steps, 4), "share_of_gpu_time": round(m / total_ms, 3),
                        "tflops": round(tf, 1) if tf else None, "frac_mfma": round(tf / pk, 4) if tf else None,
                        "gbps": round(gb, 1) if gb else None, "frac_hbm": round(gb / HBM_PEAK_GBS, 4) if gb else None})
    return roof, breakdown, classes


def rocprof_kernel(name):
    """Profiler name of a launch ('conv_wgrad3g_bf16<2,2,2,8>', 'conv_fwd_bf16<128,1,1,2,2,dma4>', ...) ->
    (kernel function, template argument list) as rocprofv3 prints them."""
    kind, _, tmpl = name.partition("<")
    t = tmpl.rstrip(">").split(",") if tmpl else []
    dt = "unsignedshort" if "bf16" in kind else ("f16_t" if "f16" in kind else "float")
    if kind.startswith("conv_wgrad3s_grp"):
        return "k_conv_wgrad3s_grp", [dt] + t
    if kind.startswith("conv_wgrad1s_grp"):                  # shared-staging kernel of the 1x1 / stride-2 layers (conv_wg1.hip)
        return "k_conv_wgrad1s_grp", [dt] + t
    if kind.startswith("fusion_gather_bwd_inv"):
        return None          # (the profile name carries no element type: no PMC row is matched for it)
    if kind.startswith("conv_wgrad3g_grp"):
        return "k_conv_wgrad3g_grp", [dt] + t
    if kind.startswith("conv_wgrad3g"):
        return "k_conv_wgrad3g", [dt] + t
    if kind.startswith("conv_wgrad3"):
        return "k_conv_wgrad3", [dt] + t
    if kind.startswith("conv_wgrad") or kind.startswith("stem_wgrad"):
        return ("k_conv_wgrad", [dt] + t) if t else None
    if t and t[0].startswith("sp"):                          # spatial-tile streaming kernel: 'conv_fwd_bf16<sp32>'
        c = t[0][2:]                                         # instantiations: <T, 32, 8, 2> (default) and <T, 64, 4, 3>
        return "k_conv3x3_sp", [dt, c, "8" if c == "32" else "4", "2" if c == "32" else "3"]
    if t and t[0].startswith("lc"):                          # loader / consumer kernel over 2-D tiles: 'conv_fwd_bf16<lc0,6x50>'
        return "k_conv3x3_lc", [dt] + {0: ["2", "5", "2", "2", "3", "440"], 1: ["1", "5", "2", "2", "4", "504"]}[int(t[0][2:])]
    if t and t[0].startswith("rs"):                          # row-sharing kernel: 'conv_fwd_bf16<rs0,9>' = tile kind 0, 9 position tiles;
        chain = "true" if any(x.startswith("x") for x in t[1:]) else "false"      # 'conv_fwd_bf16<rs2,4,x21>' = a chain of 21 layers in one launch
        l16 = "8" if (int(t[0][2:]) == 2 and chain == "false") else "0"           # loader waves: the small-M kind runs as 8 consumers + 8 loaders (option RS_L16, default)
        kind = int(t[0][2:])
        npt = int(t[1]) if len(t) > 1 and t[1].isdigit() else 99
        pf = "true" if "pf" in t[1:] else "false"             # round 6: 'conv_fwd_bf16<rs1,7,pf>' = the rotated, fragment-prefetching tap loop (11th template argument)
        if kind == 1 and npt <= 8 and chain == "false":      # <= 256 positions per workgroup: the instantiation with the pixel tile two stages ahead
            return "k_conv3x3_rs", [dt, "1", "2", "2", "4", "2", "2", "false", chain, l16, pf]
        return "k_conv3x3_rs", [dt] + {0: ["1", "5", "4", "2", "2", "1", "false"], 1: ["1", "3", "2", "4", "2", "1", "false"],
                                       2: ["1", "1", "2", "4", "6", "2", "true"]}[kind] + [chain, l16, pf]
    if kind.startswith("conv_fwd") or kind.startswith("conv_dgrad") or kind.startswith("stem_fwd"):
        tr = "true" if "dgrad" in kind else "false"
        if t and t[-1].startswith("dma"):
            return "k_conv_igemm_dma", [dt] + t[:-1] + [tr, t[-1][3:]]
        db = "true" if (t and t[-1] == "db") else "false"
        return "k_conv_igemm", [dt] + [x for x in t if x != "db"] + [tr, db]
    return None


PMC_TAG = ""          # profiles/rNNx_<tag>pmc_traffic.csv of the workload being run ("" = cfg2, the default)


def _csrc_digest():
    """sha256 over the kernel sources: tools/pmc_summary.py stores it next to a PMC summary, so that a summary collected
    on other kernels than the ones running now is recognisable."""
    import glob, hashlib
    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, PKG, "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, PKG, "csrc", "*.h"))):
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def pmc_summaries(tag=""):
    """The committed PMC summaries profiles/rNN<letters>_<tag>pmc_traffic.csv of a workload (tag "" = cfg2), oldest first: by round,
    then by the letters of the profile set (a .. z, then za, zb, ... -- the sets of a round are named in that order)."""
    import glob, re
    pat = re.compile(r"^r(\d\d)([a-z]+)_" + re.escape(tag) + r"pmc_traffic\.csv$")
    out = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_" + tag + "pmc_traffic.csv")):
        m = pat.match(os.path.basename(f))
        if m:
            out.append(((int(m.group(1)), len(m.group(2)), m.group(2)), f))
    return [f for _, f in sorted(out)]


def pmc_traffic(name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/*_pmc_traffic.csv; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).
    A bench run cannot collect PMC counters itself; returns (None, None, None) when no matching row exists.
    Third value: True when the summary was collected on different kernel sources than the ones in the tree."""
    import csv, glob
    if PMC_TAG is None:
        return None, None, None
    files = pmc_summaries(PMC_TAG)
    if not files:
        return None, None, None
    want = rocprof_kernel(name)
    if want is None:
        return None, None, None
    stale = None
    meta = files[-1][:-4] + ".meta.json"
    if os.path.exists(meta):
        stale = json.load(open(meta)).get("csrc_digest") != _csrc_digest()
    for row in csv.DictReader(open(files[-1])):
        k = row["kernel"]
        head = k.split(">(")[0] if ">(" in k else ""          # "... k_conv_wgrad3g<2, 2, 2, 8" of "...>((anonymous namespace)::WgArgs)"
        if "<" not in head:
            continue
        func = head[:head.index("<")].split("::")[-1].split()[-1]
        args = head[head.index("<") + 1:].replace(" ", "").split(",")
        if (func, args) != want:
            continue
        b = (2.0 * float(row["FETCH_SIZE_KB_per_launch_raw"]) + float(row["WRITE_SIZE_KB_per_launch"])) * 1024.0
        return round(b), os.path.basename(files[-1]), stale
    return None, None, None


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_core_counts():
    """(physical cores, logical CPUs) of the host: /proc/cpuinfo's distinct (physical id, core id) pairs, os.cpu_count()."""
    phys, cur = set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [t.strip() for t in line.split(":", 1)]
                cur[k] = v
            elif not line.strip():
                if "core id" in cur:
                    phys.add((cur.get("physical id", "0"), cur["core id"]))
                cur = {}
    except OSError:
        pass
    return (len(phys) or None), os.cpu_count()


def cpu_baseline(cfg, pool_seed):
    """The CPU restatement (oracle/, kind 'port': the reference has no camera stream / KNN / fusion code to run, SURVEY.md
    F1) on a bounded sample of the same workload: ONE cfg2 frame through the whole step on all host cores (<= 32 threads) --
    geometry (C), the brute-force KNN of the four sites (C, OpenMP over pixel rows: timed on its own, it is not what the
    baseline should be decided by), the fused model forward of oracle/model_ref.py (camera ResNet-18 + FPN, LiDAR stream,
    bilinear gather + per-neighbour MLP at each site), the reference's LossTotal restated in oracle/loss_ref.py, backward,
    Adam.  The network part runs twice and the SECOND pass is reported (the first one creates oneDNN's primitives).
    One-thread figure: LiDAR stream forward + LossTotal + backward on one thread on a 176x192 crop (1/16.7
    of the grid), second pass, scaled by the area ratio.  frames/s."""
    from oracle import geometry_ref, loss_ref, model_ref
    det = pkg("detfill")
    D = pkg("data_import_carla")
    c = copy.deepcopy(cfg)
    threads = min(os.cpu_count() or 1, 32)        # more threads than this oversubscribes oneDNN on these shapes
    torch.set_num_threads(threads)
    lim6 = (c["lidar_x_min"], c["lidar_x_max"], c["lidar_y_min"], c["lidar_y_max"], c["lidar_z_min"], c["lidar_z_max"])
    pts = det.synthetic_points(c["max_num_pc"], lim6, pool_seed)
    img = torch.from_numpy(det.synthetic_image(c["image_height"], c["image_width"], pool_seed)).unsqueeze(0)
    boxes, nb = D.synthetic_boxes(c, pool_seed)
    crt = pkg("calib").kitti_like_crt()
    t0 = time.time()
    grid, pc, uv, n, _ = geometry_ref.voxelization_projection(pts, c, crt, proj_mode="correct")
    t_geo = time.time() - t0
    g = geometry_ref.grid_constants(c)
    K = c["fusion"]["K"]
    t0 = time.time()
    maps = [[torch.from_numpy(geometry_ref.knn_bev(pc[:n], K, c["voxel_length"] // s, c["voxel_width"] // s, s, g["aff"], None))] for s in (2, 4, 8, 16)]
    t_knn = time.time() - t0
    shapes = {}
    shapes.update(model_ref.lidar_state_shapes(c))
    shapes.update(model_ref.image_state_shapes(64))
    shapes.update(model_ref.fusion_state_shapes(c, 64))
    sd = model_ref.make_state_dict(shapes)
    params = [v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k]
    opt = torch.optim.Adam(params, lr=c["learning_rate"], betas=(c["beta1"], 0.999))
    x = torch.from_numpy(grid).unsqueeze(0)
    anc = model_ref.anchors(c)
    for rep in range(2):                           # pass 0 warms oneDNN up; pass 1 is the one reported
        np.random.seed(7)
        t0 = time.time()
        pred = model_ref.forward(sd, c, x, img, torch.from_numpy(pc).unsqueeze(0), torch.from_numpy(uv).unsqueeze(0), [n], "eval",
                                 fusion={"K": K, "aff": g["aff"], "rmax": None}, knn_maps=maps)
        t_fwd = time.time() - t0
        t0 = time.time()
        loss = loss_ref.loss_total(c, boxes.unsqueeze(0), torch.tensor([nb]), pred[:, 0:4], pred[:, 4:18], anc)
        opt.zero_grad()
        loss.backward()
        opt.step()
        t_bwd = time.time() - t0
        del pred, loss
    total = t_geo + t_knn + t_fwd + t_bwd
    # one thread: LiDAR stream forward + the real loss + backward on a 176 x 192 crop of the grid, scaled by the area ratio (16.7)
    torch.set_num_threads(1)
    c1 = copy.deepcopy(c)
    ratio = (c["voxel_length"] * c["voxel_width"]) / (176.0 * 192.0)
    c1.update(dict(voxel_length=176, voxel_width=192, lidar_x_max=c["lidar_x_min"] + 176.0 / g["aff"][0], lidar_y_max=c["lidar_y_min"] + 192.0 / g["aff"][2]))
    boxes1, nb1 = D.synthetic_boxes(c1, pool_seed)
    anc1 = model_ref.anchors(c1)
    xc = x[:, :, :176, :192].contiguous()
    for rep in range(2):
        np.random.seed(7)
        t0 = time.time()
        p1 = model_ref.forward(sd, c1, xc, None, bn_mode="eval")
        l1 = loss_ref.loss_total(c1, boxes1.unsqueeze(0), torch.tensor([nb1]), p1[:, 0:4], p1[:, 4:18], anc1)
        opt.zero_grad()
        l1.backward()
        t_one = (time.time() - t0) * ratio
        del p1, l1
    torch.set_num_threads(threads)
    phys, logical = host_core_counts()
    # `cores` = the threads the baseline actually ran on (the bench contract's meaning); the host's own size next to it
    return {"value": round(1.0 / total, 4), "unit": "frames/s", "cores": threads, "threads": threads, "host_physical_cores": phys,
            "host_logical_cpus": logical, "kind": "port", "cpu": cpu_model_name(),
            "one_thread_lidar_stream_fwd_bwd_frames_per_s": round(1.0 / t_one, 4),
            "seconds": {"geometry": round(t_geo, 3), "knn_bruteforce": round(t_knn, 3), "forward": round(t_fwd, 3), "loss_backward_adam": round(t_bwd, 3)},
            "sample": "1 cfg2 frame, whole step on %d threads, second (warm) pass of the network part: C geometry %.2fs + brute-force KNN of the 4 sites "
                      "(C, OpenMP) %.2fs + fused forward (ResNet-18 camera stream, LiDAR stream, gather + per-neighbour MLP at 4 sites) %.2fs + "
                      "LossTotal + backward + Adam %.2fs; one-thread figure = LiDAR stream fwd + LossTotal + bwd on a 176x192 crop scaled by the "
                      "area ratio (%.1fs)" % (threads, t_geo, t_knn, t_fwd, t_bwd, t_one)}


def rank_envs(n, port, base=None):
    """Environment of each of the n rank processes of one node (what `python -m torch.distributed.run --nnodes=1
    --nproc-per-node n --master-addr 127.0.0.1` would export): rank r drives GPU r."""
    base = dict(os.environ if base is None else base)
    return [dict(base, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r),
                 LOCAL_WORLD_SIZE=str(n)) for r in range(n)]


def launch_ranks(n, argv):
    """Starts n rank processes of this script (one per GPU, reference train.py:24,51-56 is the single-process launcher this
    replaces) and waits for them; rank 0 inherits stdout, so its JSON line is this process's output.  Called BEFORE anything
    initialises HIP here: a process that has must not start ranks by exec, and this one only forks children.  Returns the exit
    status to leave with (the first failing rank's; the others are ended when one fails)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()              # does not initialise the GPU
    if have < n and os.environ.get("DCF_DIST_BACKEND", "nccl") == "nccl":
        print("bench.py: --gpus %d but %d GPU(s) visible (RCCL needs one device per rank; DCF_DIST_BACKEND=gloo lets ranks share a "
              "GPU for functional runs)" % (n, have), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r, env in enumerate(rank_envs(n, port)):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    status = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 1
                    for q in live:                # a rank failed: its peers would wait in a collective forever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2, help="frames per GPU (cfg2: 2)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--knn", type=int, default=3, help="neighbours per BEV pixel (cfg2: 3, cfg4: 5)")
    ap.add_argument("--image-stream", default="resnet18", help="camera trunk: resnet18 (cfg2), resnet34, resnet50 (cfg4)")
    ap.add_argument("--image", default="1242x375", help="camera frame WxH (cfg2/cfg4: 1242x375, cfg5: 1920x1080)")
    ap.add_argument("--bn-mode", default="eval", help="eval = what the reference's train.py really does (F4); train = batch statistics")
    ap.add_argument("--graphs", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="captured forward/backward HIP graphs instead of eager launches: auto (default) = at batch 1 on one GPU, where "
                    "eager launches leave the GPU waiting for the host on slower hosts (round 5, two boxes: 254.9 against 250.4 and 250.4 against 216.6 "
                    "frames/s, profiles/r05a_* / r05b_*; the compute queue is 96 % busy under replay, profiles/r05f_timeline_b1.txt); at batch >= 2 the step is kernel-bound")
    ap.add_argument("--input", default="host", choices=["host", "resident"],
                    help="where the timed steps' frames start: host (default) = host memory -> pinned staging on FrameLoader's background "
                    "thread -> H2D on a copy stream one batch ahead (SURVEY.md 8(d): the metric starts at the H2D copy of the raw frame; "
                    "VERDICT round 4 item 2); resident = raw clouds and images already in HBM.  The other mode runs as a short second leg "
                    "and is reported beside `value`")
    ap.add_argument("--from-host", action="store_true", help="same as --input host (kept from earlier rounds)")
    ap.add_argument("--no-from-host", action="store_true", help="--input resident without the second (from-host) leg")
    ap.add_argument("--no-other-leg", action="store_true", help="skip the second leg (the input mode that is not timed as `value`)")
    ap.add_argument("--loss-sampling", default="compat", help="compat = host target assignment on numpy's generator exactly like the reference's "
                    "loss.py:74-127 (default, the mode pinned to the reference); device = assignment + loss in one launch (csrc/loss.hip)")
    ap.add_argument("--loader-workers", type=int, default=0, help="DataLoader worker processes behind FrameLoader in the from-host loop "
                    "(0 = the staging thread reads and collates itself)")
    ap.add_argument("--chain", action="store_true", help="run the residual stages' 3x3 layers as chain launches (dcf_conv3x3_chain: one launch per "
                    "stage; needs the GPU to itself; level on time with the per-layer launches, DESIGN.md section 9)")
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the short resident-input legs at the other batch sizes of BASELINE.json's metric (1 / 2 / 4 / 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    env_ws = os.environ.get("WORLD_SIZE")
    if env_ws is None and args.gpus > 1:
        # `python bench.py --gpus N` typed without a launcher: this process becomes the launcher (it has not touched the GPU and
        # never will), its N children are the ranks
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if env_ws is not None and int(env_ws) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher's WORLD_SIZE is %s: one rank per GPU, the two must agree" % (args.gpus, env_ws))

    if os.environ.get("DCF_SWITCH_INTERVAL"):          # experiments: how long a thread may keep the GIL while another waits for it
        sys.setswitchinterval(float(os.environ["DCF_SWITCH_INTERVAL"]))
    train = pkg("train")
    ws = train.init_distributed()
    rank = dist.get_rank() if ws > 1 else 0
    if ws <= 1:
        torch.cuda.set_device(0)
    image_wh = tuple(int(v) for v in args.image.lower().split("x"))
    cfg = kitti_config(args.batch, args.dtype, args.points, args.knn, args.image_stream, image_wh)
    cfg["bn_mode"] = args.bn_mode
    if args.chain:
        cfg["conv_chain"] = True
    cfg["loss_sampling"] = args.loss_sampling
    global PMC_TAG
    key = (args.points, args.knn, args.image_stream, args.batch, args.dtype, args.image.lower())
    PMC_TAG = {(100000, 3, "resnet18", 2, "bf16", "1242x375"): "", (120000, 5, "resnet50", 4, "f16", "1242x375"): "cfg4_",
               (300000, 3, "resnet18", 1, "bf16", "1920x1080"): "cfg5shape_bf16_", (300000, 3, "resnet18", 1, "fp8", "1920x1080"): "cfg5_fp8_"}.get(key)
    cfg["hip_graphs"] = {"auto": "auto", "on": True, "off": False}[args.graphs]
    if args.no_from_host:
        args.input, args.no_other_leg = "resident", True
    if args.from_host:
        args.input = "host"
    args.from_host = args.input == "host"
    torch.manual_seed(0)
    np.random.seed(1234 + rank)
    trainer = train.Train(cfg)
    pkg("detfill").fill_state_dict(trainer.model)        # deterministic random-init weights (no checkpoints offline)
    pool = FramePool(cfg, n_frames=max(2 * args.batch, 4 if (ws > 1 or args.no_batch_sweep) else 8), n_points=args.points, seed0=100 * rank)

    def barrier():
        torch.cuda.synchronize()
        if ws > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if os.environ.get("DCF_BENCH_DEBUG"):
            print("[rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)

    log("model + frame pool ready")
    for s in range(args.warmup):
        train_step(trainer, pool, pool.batch(s, args.batch))
    barrier()
    log("warm-up done")
    # everything allocated so far (model, plans, frame pool) goes to the permanent generation: a full cyclic collection over it in
    # the middle of the timed region is a host pause of tens of ms that has nothing to do with the step (collection stays on)
    import gc
    gc.collect()
    gc.freeze()
    loader = None
    if args.from_host:
        loader = iter(pkg("frame_loader").FrameLoader(HostFrames(pool, args.steps + 4, args.batch), args.batch, num_workers=args.loader_workers))
        for _ in range(3):                                         # staging buffers allocated, copy engines primed, the hand-off queue in
            trainer.one_step_raw(pool.geometry, next(loader))      # steady state -- all outside the timed region
        barrier()
    # one event per step on the compute stream (recorded, never waited for inside the region): the step-to-step intervals give
    # the MEDIAN step time next to the contract's mean over the K steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    for m in marks:                      # (events are created lazily at their first record: not inside the timed region)
        m.record()
    barrier()
    t0 = time.perf_counter()
    marks[0].record()
    for s in range(args.steps):
        if loader is not None:
            trainer.one_step_raw(pool.geometry, next(loader))
        else:
            train_step(trainer, pool, pool.batch(args.warmup + s, args.batch))
        marks[s + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    raw_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    slowest = max(range(args.steps), key=lambda i: raw_ms[i])
    step_ms = sorted(raw_ms)
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    if ws > 1:
        # (a host tensor under gloo: DCF_DIST_BACKEND=gloo lets several ranks share one GPU for functional runs)
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(trainer.loss_value.item())
    frames_main = args.batch * ws * args.steps
    log("timed region done: %.3f s" % dt)

    roof, breakdown, classes, cpu, from_host, resident = None, None, None, None, None, None
    if loader is not None:
        loader.close()                                  # stops the staging thread
    if not args.no_other_leg and ws == 1:
        # the OTHER input mode of the same step, as a short second leg reported beside `value`
        n_o = min(args.steps, 20)
        if args.from_host:
            for s in range(2):
                train_step(trainer, pool, pool.batch(500 + s, args.batch))
            barrier()
            t1 = time.perf_counter()
            for s in range(n_o):
                train_step(trainer, pool, pool.batch(502 + s, args.batch))
            barrier()
            dt_o = time.perf_counter() - t1
            resident = {"value": round(args.batch * n_o / dt_o, 3), "unit": "frames/s", "ms_per_step": round(dt_o / n_o * 1e3, 3), "steps": n_o,
                        "note": "same step with the raw clouds and images already resident in HBM (no PCIe copy in the loop)"}
        else:
            fl = iter(pkg("frame_loader").FrameLoader(HostFrames(pool, n_o + 3, args.batch), args.batch))
            trainer.one_step_raw(pool.geometry, next(fl))
            trainer.one_step_raw(pool.geometry, next(fl))
            barrier()
            t1 = time.perf_counter()
            for s in range(n_o):
                trainer.one_step_raw(pool.geometry, next(fl))
            barrier()
            dt_o = time.perf_counter() - t1
            fl.close()
            from_host = {"value": round(args.batch * n_o / dt_o, 3), "unit": "frames/s", "ms_per_step": round(dt_o / n_o * 1e3, 3), "steps": n_o,
                         "note": "same step fed from host memory through FrameLoader (PCIe-inclusive)"}
    sweep = None
    if ws == 1 and not args.no_batch_sweep:
        # BASELINE.json's metric names batch 1 / 2 / 4 / 8: the other batch sizes of the same workload on the same trainer, resident
        # input, 10 timed steps each behind 4 untimed ones (batch 1 replays captured graphs under --graphs auto: they are captured there)
        sweep = {}
        for b in (1, 2, 4, 8):
            if b == args.batch or b > pool.n:
                continue
            for s in range(4):
                train_step(trainer, pool, pool.batch(700 + s, b))
            barrier()
            t1 = time.perf_counter()
            for s in range(10):
                train_step(trainer, pool, pool.batch(710 + s, b))
            barrier()
            dt_b = time.perf_counter() - t1
            sweep[str(b)] = {"frames_per_s": round(b * 10 / dt_b, 2), "ms_per_step": round(dt_b / 10 * 1e3, 3)}
        sweep[str(args.batch)] = {"frames_per_s": resident["value"] if resident else (round(frames_main / dt, 2) if not args.from_host else None),
                                  "ms_per_step": resident["ms_per_step"] if resident else (round(dt / args.steps * 1e3, 3) if not args.from_host else None)}
    if not args.no_roofline:
        # every rank runs the instrumented steps (they contain the gradient all-reduce); rank 0 reports its own
        roof, breakdown, classes = roofline_leg(trainer, pool, args.batch, 2)
    if ws > 1:
        dist.barrier()
    if rank == 0 and ws == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, 1234)
    if rank == 0:
        frames = args.batch * ws * args.steps
        out = {"metric": "frames/sec (train step) 100k-pt LiDAR + 1242x375 RGB", "value": round(frames / dt, 3), "unit": "frames/s",
               "n_gpus": ws, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "ms_per_step_median": round(median_ms, 3), "ms_per_step_min_max": [round(step_ms[0], 3), round(step_ms[-1], 3)],
               "slowest_step_index": slowest,
               "ms_first_steps": [round(marks[i].elapsed_time(marks[i + 1]), 3) for i in range(min(args.steps, 8))],
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "%s: grid 32x704x800, %d pts/frame, %s RGB, %s image stream, K=%d fusion x4 sites, "
                                      "%s-mode BN%s, batch %d/GPU" % (
                                          {(100000, 3, "resnet18", 2, "bf16", "1242x375"): "cfg2", (120000, 5, "resnet50", 4, "f16", "1242x375"): "cfg4",
                                           (300000, 3, "resnet18", 1, "fp8", "1920x1080"): "cfg5 (fp8 e4m3 forward convs with Cin>=128, bf16 storage and backward)",
                                           (300000, 3, "resnet18", 1, "bf16", "1920x1080"): "cfg5 shape at bf16"}.get(
                                              (args.points, args.knn, args.image_stream, args.batch, args.dtype, args.image.lower()), "custom"),
                                          args.points, args.image.lower(), {"resnet18": "ResNet-18", "resnet34": "ResNet-34", "resnet50": "ResNet-50"}.get(args.image_stream, args.image_stream),
                                          args.knn, args.bn_mode, " (reference F4)" if args.bn_mode == "eval" else "", args.batch),
                          "global_batch": args.batch * ws, "parallelism": "dp%d" % ws, "final_loss": round(loss, 4),
                          # SURVEY.md 8(d)'s metric starts at the H2D copy of the raw frame: by default `value` is the PCIe-inclusive rate
                          # (frames in host memory -> pinned staging on a background thread -> H2D one batch ahead: the copies of step
                          # i + 1 run under step i, so every step after the first finds its inputs in HBM); the HBM-resident rate of the
                          # same step is reported beside it
                          "input": "host memory -> pinned staging (background thread) -> H2D one batch ahead (PCIe-inclusive, SURVEY.md 8(d))" if args.from_host else "resident in HBM",
                          "from_host_frames_per_s": from_host["value"] if from_host else (round(frames / dt, 3) if args.from_host else None),
                          "from_host_ms_per_step": from_host["ms_per_step"] if from_host else (round(dt / args.steps * 1e3, 3) if args.from_host else None),
                          "resident_frames_per_s": resident["value"] if resident else (None if args.from_host else round(frames / dt, 3)),
                          "resident_ms_per_step": resident["ms_per_step"] if resident else (None if args.from_host else round(dt / args.steps * 1e3, 3)),
                          # resident-input frames/s of the same step at the batch sizes BASELINE.json's metric names (this GPU, this run)
                          "batch_sweep_frames_per_s": {k: v["frames_per_s"] for k, v in sorted(sweep.items(), key=lambda kv: int(kv[0]))} if sweep else None,
                          "batch_sweep_ms_per_step": {k: v["ms_per_step"] for k, v in sorted(sweep.items(), key=lambda kv: int(kv[0]))} if sweep else None,
                          "loss_sampling": args.loss_sampling, "conv_chain": bool(args.chain or os.environ.get("DCF_CHAIN") in ("1", "force"))},
               "roofline": roof, "cpu_baseline": cpu, "from_host": from_host, "resident": resident, "kernel_classes": classes, "kernel_breakdown": breakdown}
        print(json.dumps(out))
    if ws > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
