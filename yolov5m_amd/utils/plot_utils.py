"""Box decode -- same signatures as the reference's utils/plot_utils.py:10-54, HIP underneath.

Only the hot-path half of the reference module is mirrored (cells_to_bboxes, make_grids); the
matplotlib helpers (save_predictions, plot_image) are visualisation and out of scope.
"""
import torch

from .. import _lib


def make_grids(anchors, naxs, stride, nx=20, ny=20, i=0):
    """reference utils/plot_utils.py:42-54. Kept for API compatibility (the HIP decode kernel derives
    the grid from the cell index and never materialises it). Pure index construction, device-agnostic."""
    dev = anchors.device
    x_grid = torch.arange(nx, device=dev).repeat(ny).reshape(ny, nx)
    y_grid = torch.arange(ny, device=dev).unsqueeze(0).T.repeat(1, nx).reshape(ny, nx)
    xy_grid = torch.stack([x_grid, y_grid], dim=-1).expand(1, naxs, ny, nx, 2)
    anchor_grid = (anchors[i] * stride).reshape((1, naxs, 1, 1, 2)).expand(1, naxs, ny, nx, 2)
    return xy_grid, anchor_grid


_anchor_cache = {}


def _host_anchors(anchors):
    """(nl,na,2) anchors as host floats; cached per tensor version so decode never syncs twice."""
    key = (anchors.data_ptr(), anchors._version, tuple(anchors.shape))
    v = _anchor_cache.get(key)
    if v is None:
        v = anchors.detach().float().cpu().reshape(anchors.shape[0], -1).tolist()
        _anchor_cache.clear()
        _anchor_cache[key] = v
    return v


def cells_to_bboxes(predictions, anchors, strides, is_pred=False, to_list=True):
    """reference utils/plot_utils.py:10-40. predictions: list of (B,naxs,ny,nx,5+nc) logits
    (is_pred=True) or (B,naxs,ny,nx,6) dense targets (is_pred=False). Returns (B,N,6) rows
    [class, obj, x, y, w, h] (list of lists if to_list)."""
    L = _lib.lib()
    preds = [p if (p.is_contiguous() and p.dtype == torch.float32) else p.contiguous().float()
             for p in predictions]
    _lib.require_cuda(*preds)
    B = preds[0].shape[0]
    n_scale = [p.shape[1] * p.shape[2] * p.shape[3] for p in preds]
    N = sum(n_scale)
    out = torch.empty((B, N, 6), dtype=torch.float32, device=preds[0].device)
    hanch = _host_anchors(anchors) if is_pred else None
    off = 0
    st = _lib.stream_ptr()
    for i, p in enumerate(preds):
        _, naxs, ny, nx, nch = p.shape
        if is_pred:
            rc = L.y5m_decode_scale(_lib.ptr(p), B, naxs, ny, nx, nch - 5, _lib.float_array(hanch[i]),
                                    float(strides[i]), _lib.ptr(out), N, off, st)
        else:
            assert nch == 6
            rc = L.y5m_decode_targets_scale(_lib.ptr(p), B, naxs, ny, nx, float(strides[i]), _lib.ptr(out),
                                            N, off, st)
        _lib.check(rc, "decode")
        off += n_scale[i]
    return out.tolist() if to_list else out
