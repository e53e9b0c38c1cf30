"""Identity-keyed lazy stores for the per-process means and (cross-)kernels of a
``Measure`` (the role of ``stheno/lazy.py``): an entry is built on first access by the
first matching rule and then kept (``lazy.py:56-65``)."""

__all__ = ["LazyVector", "LazyMatrix"]


def _index(key):
    return key if isinstance(key, int) else id(key)


class _LazyStore:
    rank = 1

    def __init__(self):
        self._store = {}

    def purge(self, index):
        """Forget ``index`` (the id of a process that died) entirely: cached entries, the rules it owns and
        its membership in other rules' index sets.  CPython recycles ids, so a later process may get the
        same one -- it must not inherit the dead process's rules, nor be taken for "known" by older rules."""
        for k in [k for k in self._store if index in k]:
            del self._store[k]
        self._purge_rules(index)

    def _purge_rules(self, index):  # pragma: no cover
        pass

    def _key(self, key):
        if isinstance(key, tuple):
            return tuple(_index(k) for k in key)
        return (_index(key),) * self.rank

    def __setitem__(self, key, value):
        self._store[self._key(key)] = value

    def __getitem__(self, key):
        k = self._key(key)
        try:
            return self._store[k]
        except KeyError:
            value = self._build(k)
            self._store[k] = value
            return value

    def _build(self, k):  # pragma: no cover
        raise NotImplementedError


class LazyVector(_LazyStore):
    """Lazy vector: rules are ``(index set, builder(i))``."""

    rank = 1

    def __init__(self):
        super().__init__()
        self._rules = []

    def add_rule(self, indices, builder):
        self._rules.append((set(indices), builder))

    def _purge_rules(self, index):
        for indices, _ in self._rules:
            indices.discard(index)

    def _build(self, k):
        (i,) = k
        for indices, builder in self._rules:
            if i in indices:
                return builder(i)
        raise RuntimeError(f'Could not build value for index "{i}".')


class LazyMatrix(_LazyStore):
    """Lazy matrix: universal rules ``builder(i, j)`` are tried first, then rules that
    fix the left index, then rules that fix the right index."""

    rank = 2

    def __init__(self):
        super().__init__()
        self._rules, self._left, self._right = [], [], []

    def add_rule(self, indices, builder):
        self._rules.append((set(indices), builder))

    def add_left_rule(self, i_left, indices, builder):
        self._left.append((i_left, set(indices), builder))

    def add_right_rule(self, i_right, indices, builder):
        self._right.append((i_right, set(indices), builder))

    def _purge_rules(self, index):
        self._left = [r for r in self._left if r[0] != index]
        self._right = [r for r in self._right if r[0] != index]
        for indices, _ in self._rules:
            indices.discard(index)
        for _, indices, _ in self._left + self._right:
            indices.discard(index)

    def _build(self, k):
        i, j = k
        for indices, builder in self._rules:
            if i in indices and j in indices:
                return builder(i, j)
        for i_rule, indices, builder in self._left:
            if i == i_rule and j in indices:
                return builder(j)
        for j_rule, indices, builder in self._right:
            if j == j_rule and i in indices:
                return builder(i)
        raise RuntimeError(f"Could not build value for index {k}.")
