"""Shadows the reference's network/renderer.py for train/trainer.py:13 (`zero_thickness: False` configs: configs/shape/real/*,
configs/stage2/real/*): the same registry, built from this repo's MI355X renderers plus the reference's constructor-side dataset hook."""
from nu_nerf_amd.compat._dataset import ReferenceDatasetMixin, build_imgs_info   # noqa: F401
from nu_nerf_amd.renderer_std import NeROShapeRenderer as _Shape
from nu_nerf_amd.stage2_thick import Stage2Renderer as _Stage2


class NeROShapeRenderer(ReferenceDatasetMixin, _Shape):
    pass


class Stage2Renderer(ReferenceDatasetMixin, _Stage2):
    pass


name2renderer = {
    'shape': NeROShapeRenderer,
    'stage2': Stage2Renderer,
}
