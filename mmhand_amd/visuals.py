"""Visual outputs of the reference: tensor2im, map_to_cord, draw_pose_from_map / draw_pose_from_cords,
labelcolormap / Colorize and the 7-panel strip of MMHandModel.get_current_visuals
(util/util.py:15-21,94-191; models/MMHandModel.py:343-369).

map_to_cord runs on the device (mmh_map_to_cord, bit-exact against the reference's numpy: tests/
golden/pose.npz); the drawing itself is host-side numpy, as it is in the reference (cv2 + numpy).

Third-party arithmetic that is absent here: the reference draws with OpenCV (cv2.ellipse2Poly,
cv2.fillConvexPoly, cv2.cvtColor; cv2 is not installed and not under /root/reference).  Their
published algorithms (opencv/modules/imgproc/src/drawing.cpp) are restated below:
  * ellipse2Poly: points center + R(angle) * (a cos t, b sin t) for t = 0..360 step 1 degree with
    the float sine table, rounded half-to-even (cvRound), consecutive duplicates dropped;
  * fillConvexPoly (line_type 8, shift 0): the outline is drawn edge by edge with the 8-connected
    Bresenham line, then every scan line between the polygon's extremes is filled between the
    rounded fixed-point (16.16) edge crossings;
  * cvtColor(RGB2GRAY) of a canvas whose three channels are equal is the identity on that value.
PARITY UNPINNED for the rasterisation: no cv2 here to generate a golden strip from; the colour map
(labelcolormap) and map_to_cord are pinned by reference-generated fixtures."""
import math

import numpy as np

MISSING_VALUE = -1
# (connection, label) tables of util/util.py:24-77: the label value is the class index painted into
# the canvas and then mapped through labelcolormap(22)
BONES = [((1, 2), 2), ((2, 3), 3), ((3, 4), 4), ((5, 6), 5), ((6, 7), 6), ((7, 8), 7), ((9, 10), 8),
         ((10, 11), 9), ((11, 12), 10), ((13, 14), 11), ((14, 15), 12), ((15, 16), 13), ((17, 18), 14),
         ((18, 19), 15), ((19, 20), 16)]
PALM = [(0, 1), (1, 5), (5, 9), (9, 13), (13, 17), (17, 0)]
PALM_LABEL = 1
_SIN = np.sin(np.deg2rad(np.arange(0, 451))).astype(np.float32)      # OpenCV's SinTable (float, 0..450 deg)


def tensor2im(image_tensor, imtype=np.uint8):
    """util/util.py:15-21: first image of the batch, [-1,1] -> [0,255] (astype truncates)."""
    image_numpy = image_tensor[0].detach().cpu().float().numpy()
    if image_numpy.shape[0] == 1:
        image_numpy = np.tile(image_numpy, (3, 1, 1))
    image_numpy = (np.transpose(image_numpy, (1, 2, 0)) + 1) / 2.0 * 255.0
    return image_numpy.astype(imtype)


def labelcolormap(N):
    """util/util.py:143-160."""
    cmap = np.zeros((N, 3), dtype=np.uint8)
    for i in range(N):
        r = g = b = 0
        idx = i
        for j in range(7):
            r ^= ((idx >> 0) & 1) << (7 - j)
            g ^= ((idx >> 1) & 1) << (7 - j)
            b ^= ((idx >> 2) & 1) << (7 - j)
            idx >>= 3
        cmap[i] = (r, g, b)
    return cmap


def colorize(gray, n=22):
    """Colorize(n).add_color (util/util.py:124-141): label image [H,W] -> uint8 [H,W,3]."""
    cmap = labelcolormap(n)
    out = np.zeros(gray.shape + (3,), dtype=np.uint8)
    for label in range(n):
        out[gray == label] = cmap[label]
    return out


def map_to_cord(pose_map, threshold=0.1):
    """pose_map: device tensor [21+, H, W] (CHW) -> int array [21, 2] of (y, x), -1 when missing
    (util/util.py:94-114 on the HWC numpy array; here straight on the device, mmh_map_to_cord)."""
    from . import ops
    return ops.map_to_cord(pose_map[:21].contiguous().float(), threshold).cpu().numpy().astype(np.int64)


def _cv_round(v):
    return int(np.rint(v))          # cvRound: round half to even


def ellipse2poly(center, axes, angle, arc_start=0, arc_end=360, delta=1):
    """cv2.ellipse2Poly for integer arguments."""
    while angle < 0:
        angle += 360
    while angle > 360:
        angle -= 360
    if arc_start > arc_end:
        arc_start, arc_end = arc_end, arc_start
    while arc_start < 0:
        arc_start += 360
        arc_end += 360
    while arc_end > 360:
        arc_end -= 360
        arc_start -= 360
    if arc_end - arc_start > 360:
        arc_start, arc_end = 0, 360
    alpha, beta = float(_SIN[450 - angle]), float(_SIN[angle])       # cos, sin of the rotation
    pts, prev = [], None
    i = arc_start
    while i < arc_end + delta:
        a = min(i, arc_end)
        if a < 0:
            a += 360
        x = axes[0] * float(_SIN[450 - a])
        y = axes[1] * float(_SIN[a])
        pt = (_cv_round(center[0] + x * alpha - y * beta), _cv_round(center[1] + x * beta + y * alpha))
        if pt != prev:
            pts.append(pt)
            prev = pt
        i += delta
    if len(pts) == 1:
        pts = [tuple(center), tuple(center)]
    return np.array(pts, dtype=np.int64)


def _line(canvas, p0, p1, value):
    """8-connected Bresenham line, clipped to the canvas (cv::line, thickness 1, LINE_8)."""
    H, W = canvas.shape
    x0, y0 = int(p0[0]), int(p0[1])
    x1, y1 = int(p1[0]), int(p1[1])
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx, sy = (1 if x1 >= x0 else -1), (1 if y1 >= y0 else -1)
    if dx >= dy:
        err, y = dx // 2, y0
        for x in range(x0, x1 + sx, sx):
            if 0 <= x < W and 0 <= y < H:
                canvas[y, x] = value
            err -= dy
            if err < 0:
                y += sy
                err += dx
    else:
        err, x = dy // 2, x0
        for y in range(y0, y1 + sy, sy):
            if 0 <= x < W and 0 <= y < H:
                canvas[y, x] = value
            err -= dx
            if err < 0:
                x += sx
                err += dy


def fill_convex_poly(canvas, pts, value):
    """cv2.fillConvexPoly(canvas, pts, value) on a single-channel canvas: outline + scan-line fill."""
    pts = np.asarray(pts, dtype=np.int64).reshape(-1, 2)
    n = len(pts)
    if n == 0:
        return
    H, W = canvas.shape
    for i in range(n):
        _line(canvas, pts[i - 1], pts[i], value)
    ymin, ymax = int(pts[:, 1].min()), int(pts[:, 1].max())
    for y in range(max(ymin, 0), min(ymax, H - 1) + 1):
        xs = []
        for i in range(n):
            (xa, ya), (xb, yb) = pts[i - 1], pts[i]
            if ya == yb:
                if ya == y:
                    xs += [int(xa), int(xb)]
                continue
            if min(ya, yb) <= y <= max(ya, yb):
                # 16.16 fixed-point crossing, rounded to the nearest pixel as FillConvexPoly does
                t = (int(xa) << 16) + ((int(xb) - int(xa)) << 16) * (y - int(ya)) // (int(yb) - int(ya))
                xs.append((t + (1 << 15)) >> 16)
        if xs:
            lo, hi = max(min(xs), 0), min(max(xs), W - 1)
            if lo <= hi:
                canvas[y, lo:hi + 1] = value


def draw_pose_from_cords(pose_joints, img_size, radius=2, draw_joints=True):
    """util/util.py:164-191: the palm polygon (label 1), then one rotated ellipse (half axes
    length/2 x 8) per finger bone (labels 2..16), later shapes over earlier ones; labels -> colours
    through labelcolormap(22).  pose_joints: [21, 2] (y, x)."""
    canvas = np.zeros(tuple(img_size), dtype=np.uint8)
    palm = [(int(pose_joints[a][1]), int(pose_joints[a][0])) for a, _ in PALM]
    fill_convex_poly(canvas, palm, PALM_LABEL)
    for (a, b), label in BONES:
        y = np.array([pose_joints[a][0], pose_joints[b][0]], dtype=np.float64)
        x = np.array([pose_joints[a][1], pose_joints[b][1]], dtype=np.float64)
        m_x, m_y = x.mean(), y.mean()
        length = ((x[0] - x[1]) ** 2 + (y[0] - y[1]) ** 2) ** 0.5
        angle = math.degrees(math.atan2(y[0] - y[1], x[0] - x[1]))
        poly = ellipse2poly((int(m_x), int(m_y)), (int(length / 2), 8), int(angle), 0, 360, 1)
        fill_convex_poly(canvas, poly, label)
    return colorize(canvas, 22)


def draw_pose_from_map(pose_map, threshold=0.1, **kwargs):
    """util/util.py:116-122: pose_map [B, 21, H, W] device tensor -> uint8 [H, W, 3] of sample 0."""
    cords = map_to_cord(pose_map[0], threshold)
    return draw_pose_from_cords(cords, tuple(pose_map.shape[2:4]), **kwargs)


def visual_strip(H1, P1, D1, H2, P2, D2, fake):
    """MMHandModel.get_current_visuals (models/MMHandModel.py:343-369): H1 | P1 | D1 | H2 | P2 | D2 |
    fake, each [H, W, 3] uint8."""
    height, width = H1.shape[2], H1.shape[3]
    vis = np.zeros((height, width * 7, 3), dtype=np.uint8)
    panels = [tensor2im(H1), draw_pose_from_map(P1), tensor2im(D1), tensor2im(H2), draw_pose_from_map(P2),
              tensor2im(D2), tensor2im(fake)]
    for i, p in enumerate(panels):
        vis[:, width * i:width * (i + 1), :] = p
    return vis
