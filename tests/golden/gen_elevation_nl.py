"""
G8: convert the reference's sample file data/elevation_nl.nc (HDF5/netCDF4) to a small .npz.

Runs only in the build container, with the conda python that has h5py:

    cd /tmp && env -i PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 -B \
        /root/repo/tests/golden/gen_elevation_nl.py

Data only (node coordinates, face-node connectivity, face elevation).
"""
import os

import h5py
import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
with h5py.File("/root/reference/data/elevation_nl.nc", "r") as f:
    node_x = np.asarray(f["mesh2d_node_x"], dtype=np.float64)
    node_y = np.asarray(f["mesh2d_node_y"], dtype=np.float64)
    faces = np.asarray(f["mesh2d_face_nodes"], dtype=np.int32)
    start = int(np.asarray(f["mesh2d_face_nodes"].attrs.get("start_index", 0)).ravel()[0])
    elevation = np.asarray(f["elevation"], dtype=np.float32)
faces = faces - start
assert faces.min() == 0 and faces.max() == node_x.size - 1
np.savez_compressed(os.path.join(OUT, "g8_elevation_nl.npz"), node_x=node_x, node_y=node_y, face_nodes=faces, elevation=elevation)
print(node_x.shape, faces.shape, elevation.shape, elevation.dtype, start)
