"""Host-side box geometry of the evaluation path: rotated-box overlap for the two NMS flavours and
the precision / recall bookkeeping of the reference's test.py:110-206.

Own implementation (numpy); what has to agree with the reference are the RESULTS -- pinned by tests/golden/eval.npz,
which oracle/gen_golden.py produces by running the imported reference (its IOU.py / separation_axis_theorem.py helpers):
  * box_corners      = the corner convention behind test.py:124-134 (size = (l, w, h); l along x, h along y, w along z;
                       yaw turns about the y axis; corners 0-3 carry +h/2, 4-7 carry -h/2)
  * rotated_iou      = (3-D IoU, bird's-eye IoU) of two corner sets: the BEV footprints (x, z) are clipped against each
                       other, the 3-D overlap is the clipped area times the overlap of the two y extents
  * bev_rect / rects_overlap = the rectangle and the separating-axis test of NMS_SAT (test.py:142-175): footprints in
                       (x, y) with size[0] along the heading; touching rectangles count as overlapping
Known difference: two IDENTICAL boxes are a degenerate input of the reference's clipping (all vertices on the clip edges
of a strict half-plane test; it returns values between -11 and 3.5 for them); here the IoU of a box with itself is 1.
"""
import math

import numpy as np


def box_corners(center, size, yaw):
    l, w, h = float(size[0]), float(size[1]), float(size[2])
    c, s = math.cos(float(yaw)), math.sin(float(yaw))
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1], dtype=np.float64) * (l / 2)
    sy = np.array([1, 1, 1, 1, -1, -1, -1, -1], dtype=np.float64) * (h / 2)
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1], dtype=np.float64) * (w / 2)
    out = np.empty((8, 3), dtype=np.float64)
    out[:, 0] = c * sx + s * sz + float(center[0])
    out[:, 1] = sy + float(center[1])
    out[:, 2] = -s * sx + c * sz + float(center[2])
    return out


def _shoelace(poly):
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * abs(float(np.dot(x, np.roll(y, 1)) - np.dot(y, np.roll(x, 1))))


def _clip_convex(subject, clip):
    """Intersection polygon of `subject` with the convex, counter-clockwise polygon `clip` (half-plane by half-plane)."""
    out = [tuple(p) for p in subject]
    a = tuple(clip[-1])
    for b in clip:
        b = tuple(b)
        if not out:
            return []
        ex, ey = b[0] - a[0], b[1] - a[1]
        side = [ex * (p[1] - a[1]) - ey * (p[0] - a[0]) for p in out]        # > 0: strictly left of a->b (inside)
        nxt = []
        for i, p in enumerate(out):
            q, sq, sp = out[i - 1], side[i - 1], side[i]
            if (sp > 0) != (sq > 0):                                          # the edge q->p crosses the line a->b
                den = ex * (p[1] - q[1]) - ey * (p[0] - q[0])
                t = (ex * (a[1] - q[1]) - ey * (a[0] - q[0])) / den
                nxt.append((q[0] + t * (p[0] - q[0]), q[1] + t * (p[1] - q[1])))
            if sp > 0:
                nxt.append(p)
        out = nxt
        a = b
    return out


def rotated_iou(corners1, corners2):
    """(iou_3d, iou_bev) of two (8,3) corner arrays of box_corners()."""
    f1 = np.array([(corners1[i, 0], corners1[i, 2]) for i in (3, 2, 1, 0)])
    f2 = np.array([(corners2[i, 0], corners2[i, 2]) for i in (3, 2, 1, 0)])
    a1, a2 = _shoelace(f1), _shoelace(f2)
    inter = _clip_convex(f1, f2)
    ia = _shoelace(np.array(inter)) if len(inter) >= 3 else 0.0
    iou_bev = ia / (a1 + a2 - ia)
    top = min(corners1[0, 1], corners2[0, 1])
    bot = max(corners1[4, 1], corners2[4, 1])
    iv = ia * max(0.0, top - bot)

    def vol(c):
        return (np.linalg.norm(c[0] - c[1]) * np.linalg.norm(c[1] - c[2]) * np.linalg.norm(c[0] - c[4]))
    return iv / (vol(corners1) + vol(corners2) - iv), iou_bev


def bev_rect(center, size, yaw):
    """Footprint rectangle of NMS_SAT: 4 (x, y) vertices, size[0] along the heading, size[1] across."""
    cx, cy = float(center[0]), float(center[1])
    hl, hw = float(size[0]) / 2, float(size[1]) / 2
    c, s = math.cos(float(yaw)), math.sin(float(yaw))
    return [(cx + sl * hl * c - sw * hw * s, cy + sl * hl * s + sw * hw * c) for sl, sw in ((-1, -1), (1, -1), (1, 1), (-1, 1))]


def rects_overlap(A, B):
    """Separating-axis test of two convex polygons (lists of (x, y)); closed intervals: touching = overlapping."""
    for poly in (A, B):
        n = len(poly)
        for i in range(n):
            ex, ey = poly[(i + 1) % n][0] - poly[i][0], poly[(i + 1) % n][1] - poly[i][1]
            nx, ny = ey, -ex
            norm = math.hypot(nx, ny)
            nx, ny = nx / norm, ny / norm
            pa = [p[0] * nx + p[1] * ny for p in A]
            pb = [p[0] * nx + p[1] * ny for p in B]
            if max(pa) < min(pb) or max(pb) < min(pa):
                return False
    return True
