"""Row (b'): the reference's trainers unmodified.  `nu_nerf_amd/compat/network/` shadows network.renderer_zerothick / network.renderer
on PYTHONPATH; every other `network.*` module and `dataset.database` come from the user's NU-NeRF checkout.  Here the checkout is a
stand-in tree written to tmp_path (an interface double of dataset/database.py: parse_database_name, get_database_split, a database
with get_image / get_pose / get_K / get_depth -- no reference code), and the check runs in a fresh interpreter so that the `network`
package of this process is not involved."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE_DATABASE = '''
import numpy as np
class FakeDatabase:
    def __init__(self, name, dataset_dir):
        self.name, self.dataset_dir, self.n, self.h, self.w = name, dataset_dir, 5, 6, 8
    def get_img_ids(self): return [str(i) for i in range(self.n)]
    def get_image(self, i):
        g = np.random.Generator(np.random.PCG64(int(i)))
        return g.integers(0, 256, (self.h, self.w, 3)).astype(np.uint8)
    def get_K(self, i): return np.array([[10.0, 0, 4.0], [0, 10.0, 3.0], [0, 0, 1.0]])
    def get_pose(self, i):
        a = 0.3 * int(i)
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        return np.concatenate([R, np.array([[0.1 * int(i)], [0.2], [3.0]])], 1)
    def get_depth(self, i): return np.ones((self.h, self.w), np.float32), (np.arange(self.h * self.w).reshape(self.h, self.w) % 3 > 0)
def parse_database_name(name, dataset_dir): return FakeDatabase(name, dataset_dir)
def get_database_split(database, split_type='validation'):
    ids = database.get_img_ids()
    return ids[:1] + ids[2:], ids[1:2]
'''

DRIVER = '''
import numpy as np, torch
from network.renderer_zerothick import name2renderer, NeROShapeRenderer          # what train/trainer_zero.py:13 imports
import network.loss as refloss                                                     # any other network.* module: the checkout's
import network.renderer as std
assert refloss.MARK == "checkout" and set(name2renderer) == {"shape", "stage2"} and set(std.name2renderer) == {"shape", "stage2"}
from nu_nerf_amd.renderer import NeROShapeRenderer as Base
assert issubclass(NeROShapeRenderer, Base)
for is_nerf in (True, False):
    cfg = {"name": "x", "network": "shape", "database_name": "nerf/fake" if is_nerf else "custom/fake/8", "dataset_dir": "./datasets",
           "is_nerf": is_nerf, "train_ray_num": 16}
    torch.manual_seed(3)
    net = name2renderer[cfg["network"]](cfg)                                        # training=True is the default, as in the trainer
    assert net.database.dataset_dir == "./datasets" and net.train_num == 4 and net.test_num == 1
    assert list(net.train_ids) == ["0", "2", "3", "4"] and net.test_ids == ["1"]
    assert net.tbn == 4 * 6 * 8 and net.test_imgs_info["imgs"].shape == (1, 3, 6, 8)
    # the store equals what set_ray_store builds from the same images handed over by hand
    from nu_nerf_amd.compat._dataset import build_imgs_info, imgs_info_to_torch
    ref = Base(dict(cfg, database_name=cfg["database_name"]), training=False)
    ref.train_batch_i, ref._batch_dev, ref.test_imgs_info = 0, None, None
    torch.manual_seed(5)
    ref.set_ray_store(imgs_info_to_torch(build_imgs_info(net.database, net.train_ids, is_nerf)), device="cpu")
    key = "rays_d" if is_nerf else "dirs"
    a = torch.cat([net.train_batch[key], net.train_batch["rgbs"]], 1)
    b = torch.cat([ref.train_batch[key], ref.train_batch["rgbs"]], 1)
    assert a.shape == b.shape == (192, 6)
    sa, sb = a[torch.argsort(a @ torch.arange(1.0, 7.0))], b[torch.argsort(b @ torch.arange(1.0, 7.0))]   # (two independent shuffles of one set)
    assert torch.equal(sa, sb)
    assert float(net.train_batch["rgbs"].max()) <= 1.0 and ("masks" in net.train_batch) == is_nerf
print("compat ok")
'''


def test_unmodified_trainer_imports_resolve_and_the_constructor_loads_the_database(tmp_path):
    ck = tmp_path / "checkout"
    (ck / "network").mkdir(parents=True)
    (ck / "dataset").mkdir()
    (ck / "network" / "loss.py").write_text('MARK = "checkout"\n')
    (ck / "network" / "renderer_zerothick.py").write_text('raise RuntimeError("the reference module must be shadowed")\n')
    (ck / "network" / "renderer.py").write_text('raise RuntimeError("the reference module must be shadowed")\n')
    (ck / "dataset" / "__init__.py").write_text("")
    (ck / "dataset" / "database.py").write_text(textwrap.dedent(FAKE_DATABASE))
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "nu_nerf_amd", "compat"), ROOT, str(ck)]))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(DRIVER)], cwd=str(ck), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "compat ok" in r.stdout, r.stdout + r.stderr


def test_compat_says_what_is_missing_without_a_checkout():
    code = ("from nu_nerf_amd.compat.network.renderer_zerothick import name2renderer\n"
            "try:\n    name2renderer['shape']({'database_name': 'nerf/spherepot', 'is_nerf': True})\n"
            "except ImportError as e:\n    assert 'NU-NeRF checkout' in str(e); print('said so')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "said so" in r.stdout, r.stdout + r.stderr
