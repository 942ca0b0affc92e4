"""Writers for the files the C++ driver programs read (dsac_amd/host/frame_io.h): a scene-coordinate prediction as a .coords file, a
ground-truth pose as a 7-Scenes pose file (4 x 4 camera-to-world, metres; core/read_data.cpp:69-133 is the reader's convention),
and the optional replay files (minimal sets, refinement permutations) that make a run comparable with a recorded one."""
import os

import numpy as np


def write_coords(path, xyz, H, W, sampling=None):
    xyz = np.ascontiguousarray(np.asarray(xyz, dtype=np.float32).reshape(H * W, 3))
    with open(path, "wb") as f:
        f.write(b"DSACCRD1")
        f.write(np.array([H, W, 0 if sampling is None else 1], dtype="<i4").tobytes())
        f.write(xyz.astype("<f4").tobytes())
        if sampling is not None:
            f.write(np.ascontiguousarray(np.asarray(sampling, dtype=np.float32).reshape(H * W, 2)).astype("<f4").tobytes())


def pose_matrix_from_jp(R, t_mm, translation_txt=None):
    """The 4 x 4 a pose file must hold so that the reader ends up with the jp-convention pose (R, t [mm]): the reader computes
    (T * diag(1,-1,-1,1))^-1 after subtracting translation.txt."""
    M = np.eye(4)
    M[:3, :3] = np.asarray(R, dtype=np.float64)
    M[:3, 3] = np.asarray(t_mm, dtype=np.float64) / 1000.0
    T = np.linalg.inv(M) @ np.diag([1.0, -1.0, -1.0, 1.0])
    if translation_txt is not None:
        T[:3, 3] += np.asarray(translation_txt, dtype=np.float64)
    return T


def write_pose(path, T):
    with open(path, "w") as f:
        for row in np.asarray(T, dtype=np.float64):
            f.write(" ".join("%.9e" % v for v in row) + "\n")


def write_sets(path, sets):
    np.savetxt(path, np.asarray(sets, dtype=np.int64).reshape(-1, 4), fmt="%d")


def write_perm(path, perm):
    perm = np.ascontiguousarray(np.asarray(perm, dtype=np.int32))
    with open(path, "wb") as f:
        f.write(np.array(perm.shape, dtype="<i4").tobytes())
        f.write(perm.astype("<i4").tobytes())


def make_scene(split_dir, scene, frames):
    """frames: list of dict(name, xyz, H, W[, sampling][, pose_T][, sets][, perm]) -> <split_dir>/<scene>/{coords,poses,replay}/..."""
    base = os.path.join(split_dir, scene)
    for sub in ("coords", "poses", "replay"):
        os.makedirs(os.path.join(base, sub), exist_ok=True)
    for fr in frames:
        write_coords(os.path.join(base, "coords", fr["name"] + ".coords"), fr["xyz"], fr["H"], fr["W"], fr.get("sampling"))
        if fr.get("pose_T") is not None:
            write_pose(os.path.join(base, "poses", fr["name"] + ".pose.txt"), fr["pose_T"])
        if fr.get("sets") is not None:
            write_sets(os.path.join(base, "replay", fr["name"] + ".sets"), fr["sets"])
        if fr.get("perm") is not None:
            write_perm(os.path.join(base, "replay", fr["name"] + ".perm"), fr["perm"])
    return base
