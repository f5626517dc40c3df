import torch.nn.functional as F


class Resize:
    """torchvision.transforms.Resize for tensors == F.interpolate(bilinear, align_corners=False)."""

    def __init__(self, size, antialias=None):
        self.size = tuple(size)
        self.antialias = bool(antialias)

    def __call__(self, img):
        return F.interpolate(img, size=self.size, mode="bilinear", align_corners=False,
                             antialias=self.antialias)
