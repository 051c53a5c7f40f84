"""
``Ugrid1d``: a network of line segments -- the part of xugrid/ugrid/ugrid1d.py:65-111 and ugridbase.py the
NetworkGridder path reads: the constructor arguments ``(node_x, node_y, fill_value, edge_node_connectivity)``,
``n_node`` / ``n_edge``, ``node_coordinates``, ``edge_node_coordinates`` (ugridbase.py:611-614) and ``edge_length``
(:955-959).  UGRID IO, CRS handling, the edge kd-tree and the topology editing of the reference class are outside
the regridding hot path (out of scope).
"""
import numpy as np

from .engine import FloatDType, IntDType

FILL_VALUE = -1


class Ugrid1d:
    def __init__(self, node_x, node_y, fill_value, edge_node_connectivity, name="network1d", start_index=0):
        self.node_x = np.ascontiguousarray(node_x, dtype=FloatDType)
        self.node_y = np.ascontiguousarray(node_y, dtype=FloatDType)
        if self.node_x.ndim != 1 or self.node_x.shape != self.node_y.shape:
            raise ValueError("node_x and node_y must be 1-D arrays of equal length")
        self.fill_value = fill_value
        self.start_index = start_index
        edges = np.asarray(edge_node_connectivity)
        if edges.ndim != 2 or edges.shape[1] != 2:
            raise ValueError("edge_node_connectivity must have shape (n_edge, 2)")
        if not np.issubdtype(edges.dtype, np.integer):
            raise TypeError("edge_node_connectivity must be an integer array")
        edges = edges.astype(IntDType) - start_index
        if edges.size and (edges.min() < 0 or edges.max() >= self.node_x.size):
            raise ValueError("edge_node_connectivity refers to nodes that do not exist")
        self.edge_node_connectivity = np.ascontiguousarray(edges)
        self.name = name

    # ---- ugridbase.py:560-614, :955-959
    @property
    def n_node(self) -> int:
        return self.node_x.size

    @property
    def n_edge(self) -> int:
        return self.edge_node_connectivity.shape[0]

    @property
    def node_dimension(self):
        return f"{self.name}_nNodes"

    @property
    def edge_dimension(self):
        return f"{self.name}_nEdges"

    @property
    def core_dimension(self):
        return self.edge_dimension

    @property
    def dims(self):
        return (self.edge_dimension,)

    @property
    def node_coordinates(self):
        return np.column_stack([self.node_x, self.node_y])

    @property
    def edge_node_coordinates(self):
        """Node coordinates of every edge, shape ``(n_edge, 2, 2)``."""
        return self.node_coordinates[self.edge_node_connectivity]

    @property
    def edge_length(self):
        d = np.diff(self.edge_node_coordinates, axis=1)[:, 0, :]
        return np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])

    @property
    def bounds(self):
        return float(self.node_x.min()), float(self.node_y.min()), float(self.node_x.max()), float(self.node_y.max())

    def __eq__(self, other):
        return (
            isinstance(other, Ugrid1d)
            and np.array_equal(self.node_x, other.node_x)
            and np.array_equal(self.node_y, other.node_y)
            and np.array_equal(self.edge_node_connectivity, other.edge_node_connectivity)
        )

    __hash__ = None

    # ---- persistence (plain dict of arrays, as Ugrid2d.to_dataset)
    def to_dataset(self, prefix=None):
        name = prefix if prefix is not None else self.name
        return {
            f"{name}_node_x": self.node_x,
            f"{name}_node_y": self.node_y,
            f"{name}_edge_nodes": self.edge_node_connectivity,
        }

    @staticmethod
    def from_dataset(dataset, name):
        return Ugrid1d(
            np.asarray(dataset[f"{name}_node_x"]),
            np.asarray(dataset[f"{name}_node_y"]),
            FILL_VALUE,
            np.asarray(dataset[f"{name}_edge_nodes"]),
            name=name,
        )
