"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's train transforms (src/mngrdata.py:139-165):
RandomCrop(size, padding) -> RandomHorizontalFlip -> RandomAffine(degrees, shear, scale) -> ToTensor -> Normalize, with the
random draws passed in.  The transforms themselves live in torchvision (a dependency of the reference, imported at
src/mngrdata.py:13, not vendored and not installed here) and, for PIL images, in Pillow; restated from their published
algorithms:
  * torchvision.transforms.functional._get_inverse_affine_matrix -> `tv_inverse_affine_matrix`
  * Pillow's Image.transform(AFFINE, NEAREST) on uint8 images (Geometry.c, affine_fixed: 16.16 fixed point)
    -> `pil_affine_nearest`.
Pinned by tests/golden/affine.npz, whose expected outputs were produced by Pillow itself (tests/golden/make_golden.py)."""
import math

import numpy as np


def tv_inverse_affine_matrix(center, angle, translate, scale, shear):
    rot, sx, sy = math.radians(angle), math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def pil_fixed_coeffs(m):
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5),
            fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def pil_fixed_coeffs_abi_order(m):
    """Same six integers in matrix order (a0, a1, a2, a3, a4, a5 == pil_fixed_coeffs): the layout of
    cnk_affine_fixed_coeffs / cnk_input_pipeline_affine's affine_fix."""
    return pil_fixed_coeffs(m)


def pil_affine_nearest(img, m, fill=0):
    """img uint8 [H, W, C]; m: the 6 float coefficients handed to Image.transform(size, AFFINE, m, NEAREST)."""
    H, W = img.shape[:2]
    a0, a1, a2, a3, a4, a5 = pil_fixed_coeffs(m)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.int64)
    xin = (a2 + xs * a0 + ys * a1) >> 16
    yin = (a5 + xs * a3 + ys * a4) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    out = np.full_like(img, fill)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def random_affine(img, angle, scale, shear_x):
    """transforms.RandomAffine(degrees, shear, scale) for one draw (translate=None -> 0, shear_y = 0) on an HWC image."""
    H, W = img.shape[:2]
    return pil_affine_nearest(img, tv_inverse_affine_matrix([W * 0.5, H * 0.5], angle, [0.0, 0.0], scale, [shear_x, 0.0]))


def train_transform(images, mean, std, size, padding, crop_yx, flip, affine):
    """uint8 [N, H, W, C] -> float64 [N, C, size, size]; crop_yx [N, 2], flip [N], affine [N, 3] or None each."""
    out = []
    for n, img in enumerate(images):
        t = np.pad(img, ((padding, padding), (padding, padding), (0, 0)))                     # RandomCrop pads with 0
        oy, ox = (padding, padding) if crop_yx is None else (int(crop_yx[n][0]), int(crop_yx[n][1]))
        t = t[oy:oy + size, ox:ox + size]
        if flip is not None and flip[n]:
            t = t[:, ::-1]
        if affine is not None:
            t = random_affine(np.ascontiguousarray(t), *[float(v) for v in affine[n]])
        t = t.astype(np.float64) / 255.0                                                       # ToTensor
        out.append(((t - np.asarray(mean)) / np.asarray(std)).transpose(2, 0, 1))             # Normalize
    return np.stack(out)
