"""Literal drop-in for the reference's trainers: `run_training.py` and the YAML configs unchanged.

    PYTHONPATH=<this repo>/nu_nerf_amd/compat:<this repo>:<NU-NeRF checkout>  python run_training.py --cfg configs/shape/nerf/spherepot.yaml

`compat/network/` shadows exactly two modules of the reference's `network` package -- `renderer_zerothick` (imported by
train/trainer_zero.py:13) and `renderer` (train/trainer.py:13) -- and leaves every other `network.*` module (loss, metrics, field)
to the user's own checkout, which it appends to the package path.  The shadow modules export `name2renderer`; their classes are
this repo's renderers plus the constructor-side dataset hook of the reference (`_init_dataset`, renderer_zerothick.py:167-191,
:980-1000; renderer.py:189-196): with `training=True` on an image database name they import the USER's `dataset.database` at run
time (`parse_database_name`, `get_database_split`), stack the images as `build_imgs_info` does and hand them to `set_ray_store`.
Nothing of the reference is copied into or shipped with this repo, and nothing here is needed on a machine without the checkout.
"""
