"""Independent numpy float64 restatement of image 0.25.6's two-pass resample (sample.rs), used to
cross-check the C oracle: same index maths (evaluated in float32, as the crate does), weights and
accumulation in float64.  Rounding decisions can differ from the f32 oracle only on near-ties, so the
two must agree within 1 LSB."""
import numpy as np

F = np.float32


def lanczos3(x):
    x = np.asarray(x, dtype=np.float64)
    out = np.sinc(x) * np.sinc(x / 3.0)
    return np.where(np.abs(x) < 3.0, out, 0.0)


def gaussian(x, r):
    x = np.asarray(x, dtype=np.float64)
    return 1.0 / (np.sqrt(2.0 * np.pi) * r) * np.exp(-(x * x) / (2.0 * r * r))


def windows(in_size, out_size, support):
    ratio = F(in_size) / F(out_size)
    sratio = F(1.0) if ratio < F(1.0) else ratio
    src_support = F(support) * sratio
    res = []
    for o in range(out_size):
        c = (F(o) + F(0.5)) * ratio
        left = int(np.floor(c - src_support))
        left = min(max(left, 0), in_size - 1)
        right = int(np.ceil(c + src_support))
        right = min(max(right, left + 1), in_size)
        res.append((left, right, float(c - F(0.5)), float(sratio)))
    return res


def axis_matrix(in_size, out_size, kernel, support):
    m = np.zeros((out_size, in_size), np.float64)
    for o, (l, r, c, sr) in enumerate(windows(in_size, out_size, support)):
        w = kernel((np.arange(l, r, dtype=np.float64) - c) / sr)
        m[o, l:r] = w / w.sum()
    return m


def round_half_away(a):
    a = np.clip(a, 0.0, 255.0)
    return np.floor(a + 0.5).astype(np.uint8)


def resize_exact(img, nw, nh):
    img = img.astype(np.float64)
    mv = axis_matrix(img.shape[0], nh, lanczos3, 3.0)
    mh = axis_matrix(img.shape[1], nw, lanczos3, 3.0)
    mid = np.einsum("oy,yxc->oxc", mv, img)
    return round_half_away(np.einsum("px,oxc->opc", mh, mid))


def blur(img, sigma):
    img = img.astype(np.float64)
    k = lambda x: gaussian(x, sigma)
    mv = axis_matrix(img.shape[0], img.shape[0], k, 2.0 * sigma)
    mh = axis_matrix(img.shape[1], img.shape[1], k, 2.0 * sigma)
    mid = np.einsum("oy,yxc->oxc", mv, img)
    return round_half_away(np.einsum("px,oxc->opc", mh, mid))
