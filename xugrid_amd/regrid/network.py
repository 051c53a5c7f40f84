"""
``Network1d`` -- what ``NetworkGridder`` keeps of its source: a ``Ugrid1d`` seen as a 1-D grid whose cells are
the EDGES of the network (counterpart of xugrid/regrid/network.py).  Source data therefore has the edge dimension
as its last axis, ``size`` is the number of edges and ``length`` their lengths (the denominators the reference
would use for relative weights).
"""
from ..ugrid1d import Ugrid1d

_ACCEPTED = ("Ugrid1d", "UgridDataArray", "UgridDataset")


def _topology_of(obj) -> Ugrid1d:
    if isinstance(obj, Ugrid1d):
        return obj
    grid = getattr(obj, "grid", None)  # UgridDataArray / UgridDataset style wrappers carry their topology as .grid
    if isinstance(grid, Ugrid1d):
        return grid
    raise TypeError(f"Expected one of {set(_ACCEPTED)}, received: {type(obj).__name__}")


class Network1d:
    ndim = 1  # one spatial axis: the edges

    def __init__(self, obj):
        self.ugrid_topology = _topology_of(obj)

    size = property(lambda self: self.ugrid_topology.n_edge)
    shape = property(lambda self: (self.ugrid_topology.n_edge,))
    dims = property(lambda self: (self.ugrid_topology.edge_dimension,))
    length = property(lambda self: self.ugrid_topology.edge_length)

    def to_dataset(self, name: str):
        """The network next to the weights, like any other source grid (the reference class has no ``to_dataset``,
        so ``NetworkGridder.weights`` cannot be written there)."""
        return {**self.ugrid_topology.to_dataset(name), name + "_type": "Network1d"}
