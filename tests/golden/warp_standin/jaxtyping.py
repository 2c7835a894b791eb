"""Stand-in for ``jaxtyping`` (annotation sugar only; mpm_solver.py:10,192-196) -- TEST INFRASTRUCTURE."""


class _Ann:
    def __class_getitem__(cls, item):
        return object


class Float(_Ann):
    pass


class Int(_Ann):
    pass


class Shaped(_Ann):
    pass
