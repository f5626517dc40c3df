"""Stand-in for findiff>=0.10 (not installed; PARITY UNPINNED for the coefficient values).

Restates FinDiff(axis, h, order, acc=2).stencil(shape).data for 2-D grids: a dict keyed by the
position class per axis ('L' low edge, 'C' centre, 'H' high edge) whose values are
{(di, dj): coefficient}. acc=2 only (the reference hard-codes fd_acc=2, model.yaml:13).
"""
import itertools

_COEF = {  # (deriv order) -> class -> {offset: coeff}  (unit spacing)
    1: {"C": {-1: -0.5, 0: 0.0, 1: 0.5},
        "L": {0: -1.5, 1: 2.0, 2: -0.5},
        "H": {0: 1.5, -1: -2.0, -2: 0.5}},
    2: {"C": {-1: 1.0, 0: -2.0, 1: 1.0},
        "L": {0: 2.0, 1: -5.0, 2: 4.0, 3: -1.0},
        "H": {0: 2.0, -1: -5.0, -2: 4.0, -3: -1.0}},
}


class _Stencil:
    def __init__(self, data):
        self.data = data


class FinDiff:
    def __init__(self, *args, acc=2):
        assert acc == 2, "shim restates acc=2 only"
        if len(args) and isinstance(args[0], tuple):
            self.terms = [tuple(a) for a in args]
        else:
            self.terms = [tuple(args)]
        for t in self.terms:
            assert len(t) == 3

    def stencil(self, shape):
        ndim = len(shape)
        data = {}
        for key in itertools.product("LCH", repeat=ndim):
            per_axis = [{0: 1.0} for _ in range(ndim)]
            for axis, h, order in self.terms:
                c = _COEF[order][key[axis]]
                per_axis[axis] = {o: v / (h ** order) for o, v in c.items()}
            st = {}
            for combo in itertools.product(*[list(d.items()) for d in per_axis]):
                off = tuple(o for o, _ in combo)
                val = 1.0
                for _, v in combo:
                    val *= v
                if val != 0.0:
                    st[off] = st.get(off, 0.0) + val
            data[key] = st
        return _Stencil(data)
