"""Scan-order tables of ZigMa (host side, integer, built once at model construction).

Mirrors the reference's `utils/utils_zigzag.py`: `zigzag_path(N)` (:144-175), `hilbert_path(N)`
(:285-302, generalised Hilbert "gilbert" curve :16-130) and `reverse_permut_np` (:136-141).
Tables are numpy int64 like the reference's; `to_device_tables` turns them into the int32 device
row-index tables the HIP kernels consume (zigma_hip.h: x_row_index / z_row_index / out_row_index).
"""
import numpy as np
import torch


def reverse_permut_np(permutation):
    permutation = np.asarray(permutation, dtype=np.int64)
    reverse = np.zeros(permutation.size, dtype=np.int64)
    reverse[permutation] = np.arange(permutation.size, dtype=np.int64)
    return reverse


def zigzag_path(N):
    """8 serpentine orders of an N x N grid: for start corner TL, TR, BL, BR -> [rows-first, columns-first]."""
    idx = np.arange(N, dtype=np.int64)
    major = np.repeat(idx, N)                                    # slow index: which line
    minor = np.tile(idx, N)                                      # position along the line
    snake = np.where(major % 2 == 0, minor, N - 1 - minor)       # even lines forward, odd lines backward
    paths = []
    for start_row, start_col, dir_row, dir_col in ((0, 0, 1, 1), (0, N - 1, 1, -1), (N - 1, 0, -1, 1),
                                                   (N - 1, N - 1, -1, -1)):
        paths.append((start_row + dir_row * major) * N + start_col + dir_col * snake)   # line = row
        paths.append((start_row + dir_row * snake) * N + start_col + dir_col * major)   # line = column
    return paths


def _gilbert_cells(width, height):
    """Cells of the generalised Hilbert curve in visiting order (explicit stack instead of recursion)."""
    sgn = lambda v: (v > 0) - (v < 0)
    out = []
    stack = [(0, 0, width, 0, 0, height)] if width >= height else [(0, 0, 0, height, width, 0)]
    while stack:
        x, y, ax, ay, bx, by = stack.pop()
        w, h = abs(ax + ay), abs(bx + by)
        dax, day, dbx, dby = sgn(ax), sgn(ay), sgn(bx), sgn(by)
        if h == 1:
            out.extend((x + i * dax, y + i * day) for i in range(w))
            continue
        if w == 1:
            out.extend((x + i * dbx, y + i * dby) for i in range(h))
            continue
        ax2, ay2, bx2, by2 = ax // 2, ay // 2, bx // 2, by // 2
        if 2 * w > 3 * h:
            if abs(ax2 + ay2) % 2 and w > 2:
                ax2, ay2 = ax2 + dax, ay2 + day
            parts = [(x, y, ax2, ay2, bx, by), (x + ax2, y + ay2, ax - ax2, ay - ay2, bx, by)]
        else:
            if abs(bx2 + by2) % 2 and h > 2:
                bx2, by2 = bx2 + dbx, by2 + dby
            parts = [(x, y, bx2, by2, ax2, ay2), (x + bx2, y + by2, ax, ay, bx - bx2, by - by2),
                     (x + (ax - dax) + (bx2 - dbx), y + (ay - day) + (by2 - dby), -bx2, -by2, -(ax - ax2), -(ay - ay2))]
        stack.extend(reversed(parts))
    return out


def gilbert_order_index(N):
    order = np.zeros((N, N), dtype=np.int64)
    cells = np.asarray(_gilbert_cells(N, N), dtype=np.int64)
    order[cells[:, 0], cells[:, 1]] = np.arange(N * N, dtype=np.int64)
    return order


def hilbert_path(N=16):
    """8 variants of the order-index grid (identity / transposes of the 4 rotations), flattened.
    As in the reference these order-index grids are used directly as permutation tables."""
    res = gilbert_order_index(N)
    variants = []
    for k in range(4):
        rot = np.rot90(res, k)
        variants += [rot, np.transpose(rot)]
    return [np.ascontiguousarray(v).reshape(-1) for v in variants]


def to_device_tables(paths, device):
    return [torch.from_numpy(np.ascontiguousarray(p).astype(np.int32)).to(device) for p in paths]
