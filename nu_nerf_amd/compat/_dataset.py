"""The constructor-side dataset hook the reference's unmodified trainers rely on (renderer_zerothick.py:167-191, :980-1000;
renderer.py:189-196): database name -> images / intrinsics / poses -> the module's device-resident ray store."""
import numpy as np
import torch


def build_imgs_info(database, img_ids, is_nerf=False):
    """What the reference's module-level build_imgs_info returns (renderer_zerothick.py:20-45), from the database interface of
    dataset/database.py (`get_image` uint8 [h,w,3], `get_pose` [3,4], `get_K` [3,3], `get_depth` -> (depth, mask)): colours
    scaled to [0,1] (utils/base_utils.py color_map_forward), stacked over the images; masks for the NeRF-synthetic databases."""
    imgs = np.stack([database.get_image(i) for i in img_ids], 0).astype(np.float32) / 255.0
    info = {'imgs': imgs,
            'Ks': np.stack([database.get_K(i) for i in img_ids], 0).astype(np.float32),
            'poses': np.stack([database.get_pose(i) for i in img_ids], 0).astype(np.float32)}
    if is_nerf:
        info['masks'] = np.stack([database.get_depth(i)[1] for i in img_ids], 0)
    return info


def imgs_info_to_torch(info):
    """renderer_zerothick.py:48-56: images to float [n,3,h,w], everything else as it is."""
    out = {}
    for k, v in info.items():
        t = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32) if k.startswith('imgs') else v))
        out[k] = t.permute(0, 3, 1, 2).contiguous() if k.startswith('imgs') else t
    return out


class ReferenceDatasetMixin:
    """`_init_dataset` for image databases: the first half of the reference's method (the second half is set_ray_store).  The
    database classes come from the user's reference checkout, imported when a module is constructed with training=True."""

    def _init_dataset(self):
        super()._init_dataset()
        name = self.cfg['database_name']
        if name.startswith('synthetic'):
            return
        try:
            from dataset.database import parse_database_name, get_database_split      # the user's NU-NeRF checkout
        except ImportError as e:
            raise ImportError("nu_nerf_amd.compat needs the NU-NeRF checkout on PYTHONPATH (its dataset/database.py loads the image "
                              f"database '{name}'): PYTHONPATH=<repo>/nu_nerf_amd/compat:<repo>:<NU-NeRF checkout>") from e
        self.database = parse_database_name(name, self.cfg.get('dataset_dir'))
        train_ids, test_ids = get_database_split(self.database)
        self.train_ids, self.test_ids = np.asarray(train_ids), test_ids
        train = imgs_info_to_torch(build_imgs_info(self.database, self.train_ids, self.is_nerf))
        test = imgs_info_to_torch(build_imgs_info(self.database, self.test_ids, self.is_nerf))
        _, _, h, w = train['imgs'].shape
        print(f'training size {h} {w} ...')
        # built on the CPU like the reference's (the parameters are not on the GPU yet: the trainer calls .cuda() on the finished
        # module); the first train_step moves the store to the module's device once and slices it there from then on
        self.set_ray_store(train, test, device='cpu')
        self.train_num, self.test_num = len(train_ids), len(test_ids)
