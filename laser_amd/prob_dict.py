"""Dictionary view over a dense probability tensor.

The reference returns Python dicts keyed by tuples whose values are 0-dim
tensors sliced out of the dense [keywords, crops] result (reference:
laser/models/llava_clip_model_v3.py:290-339, 464-474).  Building 10^5 such
entries eagerly costs more than the whole GPU pipeline, so the scoring module
returns this Mapping instead: same keys, same iteration order, same values
(0-dim views of the dense tensor), materialised on first use.  Dense consumers
read `.dense`, `.row_names` and `.col_keys` directly.
"""
from __future__ import annotations

from collections.abc import Mapping
from typing import Callable, Hashable, List, Sequence, Tuple

import torch


class ProbDict(Mapping):
    def __init__(self, dense: torch.Tensor, rows: Sequence[Tuple[int, str]], cols: Sequence[Tuple[int, Hashable]],
                 make_key: Callable[[Hashable, str], Hashable], order: str):
        """dense: [R, C]; rows: (row index, name) of the non-dummy keywords in
        reference order; cols: (column index, column key part) of the columns that
        take part; order: 'row_major' (names outer, columns inner) or 'col_major'."""
        self.dense = dense
        self.row_names = list(rows)
        self.col_keys = list(cols)
        self._make_key = make_key
        self._order = order
        self._pos = None  # key -> flat index into dense, insertion-ordered, last write wins

    # -- lazy index ----------------------------------------------------------
    def _index(self):
        if self._pos is None:
            ncol = self.dense.shape[1] if self.dense.dim() == 2 else 0
            pos = {}
            mk = self._make_key
            if self._order == "row_major":
                for r, name in self.row_names:
                    base = r * ncol
                    for c, ck in self.col_keys:
                        pos[mk(ck, name)] = base + c
            else:
                for c, ck in self.col_keys:
                    for r, name in self.row_names:
                        pos[mk(ck, name)] = r * ncol + c
            self._pos = pos
        return self._pos

    def _flat_values(self) -> torch.Tensor:
        pos = self._index()
        if not pos:
            return self.dense.new_zeros((0,))
        idx = torch.tensor(list(pos.values()), dtype=torch.long, device=self.dense.device)
        return self.dense.reshape(-1)[idx]

    # -- Mapping -------------------------------------------------------------
    def __getitem__(self, key):
        return self.dense.reshape(-1)[self._index()[key]]

    def __iter__(self):
        return iter(self._index())

    def __len__(self):
        return len(self._index())

    def __contains__(self, key):
        return key in self._index()

    def keys(self):
        return self._index().keys()

    def values(self):
        return list(self._flat_values().unbind(0))

    def items(self):
        return list(zip(self._index().keys(), self._flat_values().unbind(0)))

    # -- conveniences --------------------------------------------------------
    def values_tensor(self) -> torch.Tensor:
        """1-D tensor of the values in key order (one gather)."""
        return self._flat_values()

    def to_dict(self) -> dict:
        return dict(self.items())

    def __reduce__(self):
        return (dict, (list(zip(self._index().keys(), self._flat_values().detach().cpu().unbind(0))),))

    def __repr__(self):
        return f"ProbDict({len(self)} entries over dense {tuple(self.dense.shape)})"
