"""``Network1d``: the regridder-side adapter of a ``Ugrid1d`` -- xugrid/regrid/network.py:4-36."""
from ..ugrid1d import Ugrid1d


class Network1d:
    def __init__(self, obj):
        if isinstance(obj, Ugrid1d):
            self.ugrid_topology = obj
        elif hasattr(obj, "grid") and isinstance(obj.grid, Ugrid1d):
            self.ugrid_topology = obj.grid  # UgridDataArray-like wrapper
        else:
            options = {"Ugrid1d", "UgridDataArray", "UgridDataset"}
            raise TypeError(f"Expected one of {options}, received: {type(obj).__name__}")

    @property
    def ndim(self):
        return 1

    @property
    def dims(self):
        return (self.ugrid_topology.edge_dimension,)

    @property
    def shape(self):
        return (self.ugrid_topology.n_edge,)

    @property
    def size(self):
        return self.ugrid_topology.n_edge

    @property
    def length(self):
        return self.ugrid_topology.edge_length

    # (the reference class has no to_dataset, so ``NetworkGridder.weights`` cannot be written there; here the
    # network is stored next to the weights like any other source grid)
    def to_dataset(self, name: str):
        ds = self.ugrid_topology.to_dataset(name)
        ds[name + "_type"] = "Network1d"
        return ds
