"""``compressai.datasets.ImageFolder`` -- the stereo loader of the HESIC scripts (reference:
compressai/datasets/utils.py:68-214).  CPU-side data preparation in front of the accelerated path:

* ``root/<split>/left/*`` and ``root/<split>/right/*`` paired by file name, one random crop position shared by both views;
* the homography ``H`` (left pixel -> right pixel, 3x3 fp32) the model's ``warp_perspective`` takes.  The reference computes it
  per item with OpenCV-contrib SURF + BFMatcher + RANSAC (``get_H``, utils.py:30-66) -- a non-free module this image does not
  have (and the reference pins opencv-contrib 3.4.2.17 for it).  Here it comes from, in this order: a user callable
  ``homography(img1, img2) -> 3x3 | None``; a precomputed sidecar ``root/<split>/H/<stem>.{npy,txt}`` holding the 3x3 matrix
  of the FULL images (re-expressed in crop coordinates: both views are cropped at the same offset t, so H_crop = T(-t) H T(t));
  OpenCV's SURF route when ``cv2.xfeatures2d`` exists.  No source -> the item is ``(img1, img2)`` only, which is also what the
  reference returns when RANSAC fails (utils.py:189-197);
* the HomographyNet inputs of the ``_real`` scripts (utils.py:161-186): both crops resized to 256 x 256, normalised with the
  channel-mean of the ImageNet statistics, averaged to grey, one random 128 x 128 window with its corner coordinates.
"""
import glob
import os
import random
from pathlib import Path

import numpy as np
import torch

MEAN = torch.tensor([0.485, 0.456, 0.406]).mean().unsqueeze(0)
STD = torch.tensor([0.229, 0.224, 0.225]).mean().unsqueeze(0)


def _read_rgb(path):
    """(H, W, 3) uint8 RGB array (what cv2.imread + BGR2RGB gives the reference)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.array(im.convert("RGB"))          # a writable copy (torch.from_numpy wants one)


def _resize_bilinear(img, size):
    """cv2.resize(img, (size, size)) of the reference (INTER_LINEAR: half-pixel centres, no anti-aliasing) on a uint8 HWC
    array; OpenCV works in 11-bit fixed point, this is the float form rounded to uint8 (equal to +-1 grey level)."""
    t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).unsqueeze(0).float()
    t = torch.nn.functional.interpolate(t, size=(size, size), mode="bilinear", align_corners=False, antialias=False)
    return t.round().clamp(0, 255).to(torch.uint8)[0].permute(1, 2, 0).numpy()


def surf_ransac_homography(img1, img2):
    """The reference's ``get_H`` (utils.py:30-66) where OpenCV-contrib is installed: SURF keypoints, 2-NN brute-force matching
    with the 0.7 ratio test, ``cv2.findHomography(RANSAC, 5.0)``.  Returns a 3x3 float32 array or None."""
    import cv2
    surf = cv2.xfeatures2d.SURF_create()
    kp1, des1 = surf.detectAndCompute(img1, None)
    kp2, des2 = surf.detectAndCompute(img2, None)
    if des1 is None or des2 is None:
        return None
    good = [m for m, n in (p for p in cv2.BFMatcher().knnMatch(des1, des2, k=2) if len(p) == 2) if m.distance < 0.7 * n.distance]
    if len(good) < 4:
        return None
    src = np.float32([kp1[m.queryIdx].pt for m in good]).reshape(-1, 1, 2)
    dst = np.float32([kp2[m.trainIdx].pt for m in good]).reshape(-1, 1, 2)
    H, _ = cv2.findHomography(src, dst, cv2.RANSAC, 5.0)
    return None if H is None else H.astype(np.float32)


def _have_surf():
    try:
        import cv2
        return hasattr(cv2, "xfeatures2d") and hasattr(cv2.xfeatures2d, "SURF_create")
    except Exception:
        return False


class ImageFolder(torch.utils.data.Dataset):
    """Stereo image folder ``root/<split>/{left,right}/`` (same constructor as the reference's, plus ``homography``).

    Item: ``(img1, img2, H, homo_img1, homo_img2, corners)`` -- with ``need_file_name`` the file name follows ``H`` -- where
    img1 / img2 are the paired crops (``transform`` applied: the scripts pass ToTensor), ``H`` the crop-frame homography,
    homo_img* the normalised grey (1, 128, 128) HomographyNet windows and ``corners`` their (4, 2) corner coordinates in the
    256 x 256 frame (x, y; clockwise from the top-left)."""

    def __init__(self, root, transform=None, patch_size=(256, 256), split="train", need_file_name=False, homography=None):
        splitdir = Path(root) / split
        if not splitdir.is_dir():
            raise RuntimeError(f'Invalid directory "{root}"')
        self.left_list = sorted(glob.glob(os.path.join(splitdir / "left", "*")))
        self.right_list = sorted(glob.glob(os.path.join(splitdir / "right", "*")))
        if len(self.left_list) != len(self.right_list):
            raise RuntimeError(f"{splitdir}: {len(self.left_list)} left images but {len(self.right_list)} right images")
        self.patch_size = tuple(patch_size)
        self.transform = transform
        self.need_file_name = need_file_name
        self.homography = homography
        self.sidecar_dir = splitdir / "H"
        self.homopic_size, self.homopatch_size, self.rho = 256, 128, 45

    def __len__(self):
        return len(self.left_list)

    # ---- homography of a pair, in the coordinates of the crop that starts at (x0, y0)
    def _sidecar(self, stem):
        for ext in (".npy", ".txt"):
            f = self.sidecar_dir / (stem + ext)
            if f.is_file():
                H = np.load(f) if ext == ".npy" else np.loadtxt(f)
                return np.asarray(H, dtype=np.float64).reshape(3, 3)
        return None

    def _pair_homography(self, index, img1, img2, x0, y0):
        if self.homography is not None:
            H = self.homography(img1, img2)
            return None if H is None else torch.as_tensor(np.asarray(H, dtype=np.float32)).reshape(3, 3)
        full = self._sidecar(Path(self.left_list[index]).stem)
        if full is not None:
            t_in = np.array([[1, 0, x0], [0, 1, y0], [0, 0, 1]], dtype=np.float64)       # crop pixel -> full-image pixel (left)
            t_out = np.array([[1, 0, -x0], [0, 1, -y0], [0, 0, 1]], dtype=np.float64)    # full-image pixel -> crop pixel (right)
            H = t_out @ full @ t_in
            return torch.from_numpy((H / H[2, 2]).astype(np.float32))
        if _have_surf():
            H = surf_ransac_homography(img1, img2)
            return None if H is None else torch.from_numpy(H)
        return None

    def _homonet_inputs(self, img1, img2):
        S, P, rho = self.homopic_size, self.homopatch_size, self.rho
        greys = []
        for im in (img1, img2):
            t = torch.from_numpy(_resize_bilinear(im, S)).permute(2, 0, 1).float().div_(255.0)        # ToTensor
            t = (t - MEAN.view(1, 1, 1)) / STD.view(1, 1, 1)                                           # Normalize(mean, std)
            greys.append(t.mean(dim=0, keepdim=True))
        if S - rho - P >= rho:
            x, y = random.randint(rho, S - rho - P), random.randint(rho, S - rho - P)
        else:
            x = y = 0
        corners = torch.tensor([[x, y], [x + P, y], [x + P, y + P], [x, y + P]], dtype=torch.float32)
        return greys[0][:, y:y + P, x:x + P], greys[1][:, y:y + P, x:x + P], corners

    def __getitem__(self, index):
        left, right = self.left_list[index], self.right_list[index]
        if os.path.basename(left) != os.path.basename(right):
            raise ValueError("cannot compare pictures.")
        img1, img2 = _read_rgb(left), _read_rgb(right)
        if img1.shape != img2.shape:
            raise ValueError(f"{os.path.basename(left)}: the two views differ in size ({img1.shape} vs {img2.shape})")
        Himg, Wimg, _ = img1.shape
        ph, pw = self.patch_size
        if ph > Himg or pw > Wimg:
            raise ValueError(f"{os.path.basename(left)}: patch {self.patch_size} larger than the image ({Himg}, {Wimg})")
        if ph == Himg:                                   # the reference's rule (utils.py:147-152): full height -> no offset at all
            y0 = x0 = 0
        else:
            y0, x0 = random.randint(0, Himg - ph - 1), random.randint(0, max(Wimg - pw - 1, 0))
        img1, img2 = img1[y0:y0 + ph, x0:x0 + pw], img2[y0:y0 + ph, x0:x0 + pw]
        H = self._pair_homography(index, img1, img2, x0, y0)
        homo1, homo2, corners = self._homonet_inputs(img1, img2)
        a, b = (self.transform(img1), self.transform(img2)) if self.transform else (img1, img2)
        if H is None:
            return a, b
        if self.need_file_name:
            return a, b, H, os.path.basename(left), homo1, homo2, corners
        return a, b, H, homo1, homo2, corners


def to_tensor(img):
    """torchvision.transforms.ToTensor for a uint8 HWC array (the one transform the HESIC scripts pass): CHW float in [0, 1]."""
    return torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div_(255.0)


__all__ = ["ImageFolder", "to_tensor", "surf_ransac_homography"]
