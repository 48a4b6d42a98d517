"""Box / polar-coordinate helpers of the decoder hot path (device-agnostic tensor math).

Counterparts of ``models/bbox/utils.py:26-106`` and ``models/utils.py:86-101`` of the reference,
same names and conventions.  Query box vector (10): [theta (turns), d (/65 m), z (norm), log w,
log l, log h, sin, cos, vx, vy]."""
import math

import torch

_TWO_PI = 2 * math.pi
_CONSTS = {}


def const_tensor(like, values):
    """Small constant vector on ``like``'s device / dtype, uploaded once.  (``like.new_tensor(list)`` is a
    blocking pageable host-to-device copy: on the GPU it stalls the host until the stream has drained, once per
    call, and cannot be captured in a HIP graph.)"""
    key = (str(like.device), like.dtype, tuple(float(v) for v in values))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(list(key[2]), device=like.device, dtype=like.dtype)
    return t


def decode_bbox(bboxes, pc_range=None):
    """bbox/utils.py:66-80 -> [x, y, z, w, l, h, yaw, (vx, vy)]"""
    xyz = bboxes[..., 0:3]
    if pc_range is not None:
        lo = const_tensor(bboxes, pc_range[0:3])
        span = const_tensor(bboxes, [pc_range[3] - pc_range[0], pc_range[4] - pc_range[1], pc_range[5] - pc_range[2]])
        xyz = xyz * span + lo
    parts = [xyz, bboxes[..., 3:6].exp(), torch.atan2(bboxes[..., 6:7], bboxes[..., 7:8])]
    if bboxes.shape[-1] > 8:
        parts.append(bboxes[..., 8:10])
    return torch.cat(parts, dim=-1)


def denormalize_bbox(normalized_bboxes):
    """bbox/utils.py:26-46 (input order cx, cy, w, l, cz, h, sin, cos, vx, vy)"""
    nb = normalized_bboxes
    rot = torch.atan2(nb[..., 6:7], nb[..., 7:8])
    parts = [nb[..., 0:1], nb[..., 1:2], nb[..., 4:5], nb[..., 2:3].exp(), nb[..., 3:4].exp(),
             nb[..., 5:6].exp(), rot]
    if nb.size(-1) > 8:
        parts += [nb[..., 8:9], nb[..., 9:10]]
    return torch.cat(parts, dim=-1)


def theta_d2xy_coods(theta_d_coords, map_size=102.4, r=65.0):
    """bbox/utils.py:82-90: polar -> normalised xy, clamped to [0,1]; other dims pass through."""
    center = map_size / 2
    ang = theta_d_coords[..., 0:1] * _TWO_PI
    rad = theta_d_coords[..., 1:2] * r
    xy = torch.cat([(center + rad * torch.cos(ang)) / map_size,
                    (center + rad * torch.sin(ang)) / map_size], dim=-1).clamp(min=0, max=1)
    return torch.cat([xy, theta_d_coords[..., 2:]], dim=-1)


def xy2theta_d_coods(xy_coords_norm, map_size=102.4, r=65.0, norm=True):
    """bbox/utils.py:93-106"""
    if norm:
        center = map_size / 2
        dx = xy_coords_norm[..., 0:1] * map_size - center
        dy = xy_coords_norm[..., 1:2] * map_size - center
        dist = torch.sqrt(dx ** 2 + dy ** 2) / r
        theta = ((torch.atan2(dy, dx) + _TWO_PI) % _TWO_PI) / _TWO_PI
    else:
        dx, dy = xy_coords_norm[..., 0:1], xy_coords_norm[..., 1:2]
        dist = torch.sqrt(dx ** 2 + dy ** 2)
        theta = (torch.atan2(dy, dx) + _TWO_PI) % _TWO_PI
    return torch.cat([theta, dist, xy_coords_norm[..., 2:]], dim=-1)


def inverse_sigmoid(x, eps=1e-5):
    """models/utils.py:86-101"""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))
