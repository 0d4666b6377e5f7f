"""Camera calibration constants (host-side, numpy only).

CRT = (K . [R|0])^T as a [4,3] float32 matrix, the layout of CarlaDataset.CRT_tensor
(/root/reference/data_import_carla.py:31-34).
"""
import numpy as np


def euler_zyz_rotation(v):
    """ZYZ euler angles -> rotation matrix: what data_import_carla.py:185-186 asks of
    numpy-quaternion (from_euler_angles + as_rotation_matrix).  That package is not pinned
    by the reference, so this restatement is "parity unpinned" (DESIGN.md)."""
    a, b, g = v
    q = np.array([np.cos(b / 2) * np.cos((a + g) / 2), -np.sin(b / 2) * np.sin((a - g) / 2),
                  np.sin(b / 2) * np.cos((a - g) / 2), np.cos(b / 2) * np.sin((a + g) / 2)])
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def crt_from(K, R):
    RT = np.concatenate((np.asarray(R, dtype=np.float64), np.zeros((3, 1))), axis=-1)
    return np.ascontiguousarray(np.matmul(np.asarray(K, dtype=np.float64), RT).T.astype(np.float32))


def carla_crt():
    """data_import_carla.py:180-194: CARLA intrinsics and the euler-difference extrinsic."""
    v_lidar = np.array([-1.57079633, 3.12042851, -1.57079633])
    v_cam = np.array([-3.13498819, 1.59196951, 1.56942932])
    R = euler_zyz_rotation(v_cam - v_lidar)
    K = np.array([[268.51188197672957, 0.0, 320.0], [0.0, 268.51188197672957, 240.0], [0.0, 0.0, 1.0]])
    return crt_from(K, R)


# lidar (x forward, y left, z up) -> camera (z forward, x right, y down)
R_LIDAR_TO_CAM = np.array([[0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.0]])


def kitti_like_crt():
    """KITTI-like intrinsics for the 1242x375 benchmark frames."""
    K = np.array([[721.5377, 0.0, 609.5593], [0.0, 721.5377, 172.854], [0.0, 0.0, 1.0]])
    return crt_from(K, R_LIDAR_TO_CAM)


def hd_crt():
    """1920x1080 stress configuration."""
    K = np.array([[1000.0, 0.0, 960.0], [0.0, 1000.0, 540.0], [0.0, 0.0, 1.0]])
    return crt_from(K, R_LIDAR_TO_CAM)
