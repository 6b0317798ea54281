"""``compressai.datasets.ImageFolder`` -- name kept importable (ywz/mywork/newnet1.py:27).

The reference's stereo loader (compressai/datasets/utils.py:68-214: PNG decode, paired crops,
SURF + RANSAC homography via OpenCV-contrib) is CPU data preparation outside the accelerated path
(SURVEY.md 8f rank 4); benchmarks and tests use ``hesic_amd.synthetic`` pairs instead."""


class ImageFolder:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "compressai.datasets.ImageFolder (OpenCV SURF/RANSAC stereo loader) is outside the MI355X hot path; "
            "use hesic_amd.synthetic.stereo_batch or your own Dataset yielding (x1, x2, H)")


__all__ = ["ImageFolder"]
