"""Genotype container and decoder (drop-in for the reference's genotype.py:6-45); host-side numpy."""
from collections import namedtuple

import numpy as np

from .prim_ops import DownOps, NormOps, UpOps

Genotype = namedtuple("Genotype", ["down", "up"])


class GenoParser:
    def __init__(self, n_nodes):
        self.n_nodes = n_nodes

    def parse(self, alpha1, alpha2, downward=True):
        """alpha1 / alpha2: softmaxed (n_edges, n_prims) matrices for stride-1 / stride-2 edges.
        Per edge keep the arg-max primitive; per node keep the two best-scoring incoming edges,
        stride-2 scores rescaled by len(primitive list)/len(NormOps) (genotype.py:28-45)."""
        alpha1, alpha2 = np.asarray(alpha1), np.asarray(alpha2)
        strided_names = DownOps if downward else UpOps
        picked = []
        e = 0
        for n_in in range(2, 2 + self.n_nodes):
            scored = []
            for edge in range(n_in):
                is_strided = (edge < 2) if downward else (edge == 1)
                if is_strided:
                    j = int(np.argmax(alpha2[e]))
                    # the reference's order of operations, in the matrix's own dtype: (a * n_strided) / n_normal -- on float32
                    # scores a precomputed ratio differs by 1 ulp for half of all values, which decides a near-tie (genotype.py:35,38)
                    scored.append((alpha2[e][j] * len(strided_names) / len(NormOps), strided_names[j], edge))
                else:
                    j = int(np.argmax(alpha1[e]))
                    scored.append((alpha1[e][j], NormOps[j], edge))
                e += 1
            scored.sort()
            picked.extend((name, edge) for _, name, edge in scored[-2:])
        return picked
