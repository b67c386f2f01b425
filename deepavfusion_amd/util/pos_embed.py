"""Fixed sin-cos position tables (host-side constants; reference util/pos_embed.py:16-90)."""
import numpy as np


def _sincos_1d(dim, pos):
    # util/pos_embed.py:72-90: [sin | cos](pos / 10000^(i / (dim/2)))
    assert dim % 2 == 0
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float32) / (dim / 2.0))
    ang = np.outer(np.asarray(pos, dtype=np.float32).reshape(-1), omega)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """[gh*gw (+1), embed_dim]; first half of the channels encodes the W coordinate
    (the reference meshgrids (w, h), util/pos_embed.py:51)."""
    if isinstance(grid_size, int):
        grid_size = (grid_size, grid_size)
    gh, gw = grid_size
    xs, ys = np.meshgrid(np.arange(gw, dtype=np.float32), np.arange(gh, dtype=np.float32))
    emb = np.concatenate([_sincos_1d(embed_dim // 2, xs), _sincos_1d(embed_dim // 2, ys)], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([1, embed_dim]), emb], axis=0)
    return emb


def get_3d_sincos_pos_embed(embed_dim, grid_size, cls_token=0, thw_props=(2, 1, 1)):
    """[gt*gh*gw (+cls), embed_dim] (util/pos_embed.py:16-40): channels split t : h : w = thw_props.  The reference
    meshgrids (t, w, h) with 'ij' indexing and then views the three coordinate volumes as [gt, gh, gw]; the same
    reinterpretation is done here so that non-square grids get the identical table."""
    gt, gh, gw = grid_size
    h_dim = int(embed_dim * (thw_props[1] / float(sum(thw_props))))
    w_dim = int(embed_dim * (thw_props[2] / float(sum(thw_props))))
    t_dim = embed_dim - h_dim - w_dim
    at, aw, ah = np.arange(gt, dtype=np.float32), np.arange(gw, dtype=np.float32), np.arange(gh, dtype=np.float32)
    ct = np.broadcast_to(at[:, None, None], (gt, gw, gh)).reshape(-1)       # coordinate volumes in (t, w, h) order ...
    cw = np.broadcast_to(aw[None, :, None], (gt, gw, gh)).reshape(-1)       # ... read back flat, i.e. as [gt, gh, gw]
    ch = np.broadcast_to(ah[None, None, :], (gt, gw, gh)).reshape(-1)
    emb = np.concatenate([_sincos_1d(t_dim, ct), _sincos_1d(h_dim, cw), _sincos_1d(w_dim, ch)], axis=1)
    if cls_token:
        emb = np.concatenate([np.zeros([int(cls_token), embed_dim]), emb], axis=0)
    return emb
