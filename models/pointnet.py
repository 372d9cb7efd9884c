"""Drop-in for the reference's ``models/pointnet.py``: same import path, same class name and call
signature (reference models/pointnet.py:8-99), backed by the MI355X kernels in ``scream_amd``.

    from models.pointnet import PointTransformer
    net = PointTransformer(d_model=256); net.to("cuda:0"); net.load_state_dict(torch.load(...)); net.eval()
    src_, imgs, transform = net(src, tgt, src_center, s, False, get_transform, filter)
"""
from scream_amd.model import DEMTransformer, PointTransformer  # noqa: F401

__all__ = ["PointTransformer", "DEMTransformer"]
