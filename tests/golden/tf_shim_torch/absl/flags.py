"""absl.flags stand-in: DEFINE_* record defaults on a FLAGS object whose attributes the caller may overwrite."""


class _Flags:
    pass


FLAGS = _Flags()


def _define(name, default, help=None, **_):  # noqa: A002
    setattr(FLAGS, name, default)


DEFINE_string = DEFINE_integer = DEFINE_float = DEFINE_boolean = DEFINE_bool = DEFINE_enum = _define
