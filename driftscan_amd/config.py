"""Configuration properties, a small stand-in for the part of ``caput.config`` the
reference's operator classes are written against (drift/core/beamtransfer.py:186-195,
kltransform.py:164-177, telescope.py:211-243): class-level ``Property`` descriptors
that can be filled from a (YAML) dictionary with ``read_config`` / ``from_config``.
"""


class Property(object):
    """A typed, defaulted attribute that ``Reader.read_config`` can set from a dict.

    proptype : callable applied to incoming values (None passes through)
    default  : value returned until something is set
    key      : dictionary key to read from (defaults to the attribute name)
    """

    def __init__(self, default=None, proptype=None, key=None):
        self.default = default
        self.proptype = proptype
        self.key = key
        self.name = None

    def __set_name__(self, owner, name):
        self.name = name
        if self.key is None:
            self.key = name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        return obj.__dict__.get("_cfg_" + self.name, self.default)

    def __set__(self, obj, value):
        if value is not None and self.proptype is not None:
            value = self.proptype(value)
        obj.__dict__["_cfg_" + self.name] = value


def enum(options, default=None):
    options = list(options)

    def check(v):
        if v not in options:
            raise ValueError("%r is not one of %r" % (v, options))
        return v

    return Property(default=default, proptype=check)


def list_type(type_=None, length=None, maxlength=None, default=None):
    def conv(v):
        v = list(v)
        if length is not None and len(v) != length:
            raise ValueError("list must have length %d" % length)
        if maxlength is not None and len(v) > maxlength:
            raise ValueError("list must have at most %d entries" % maxlength)
        return [type_(x) for x in v] if type_ is not None else v

    return Property(default=default, proptype=conv)


def _yesno(v):
    if isinstance(v, str):
        return v.strip().lower() in ("yes", "true", "y", "1", "on")
    return bool(v)


truthy = _yesno


class Reader(object):
    """Mixin: ``obj.read_config(dict)`` sets every Property whose key is present;
    ``Cls.from_config(dict, *args)`` constructs then configures."""

    @classmethod
    def _properties(cls):
        seen = {}
        for klass in reversed(cls.__mro__):
            for name, val in vars(klass).items():
                if isinstance(val, Property):
                    seen[name] = val
        return seen

    def read_config(self, cfg):
        for name, prop in self._properties().items():
            if prop.key in cfg:
                setattr(self, name, cfg[prop.key])
        self._finalise_config()

    def _finalise_config(self):
        pass

    @classmethod
    def from_config(cls, cfg, *args, **kwargs):
        obj = cls(*args, **kwargs)
        obj.read_config(cfg)
        return obj
