"""Shadows the reference's network/renderer_zerothick.py for train/trainer_zero.py:13 (`zero_thickness: True` configs): the same
registry, built from this repo's MI355X renderers plus the reference's constructor-side dataset hook."""
from nu_nerf_amd.compat._dataset import ReferenceDatasetMixin, build_imgs_info   # noqa: F401  (build_imgs_info: module-level name of the reference)
from nu_nerf_amd.renderer import NeROShapeRenderer as _Shape
from nu_nerf_amd.stage2 import Stage2Renderer as _Stage2


class NeROShapeRenderer(ReferenceDatasetMixin, _Shape):
    pass


class Stage2Renderer(ReferenceDatasetMixin, _Stage2):
    pass


name2renderer = {
    'shape': NeROShapeRenderer,
    'stage2': Stage2Renderer,
}
