"""Minimal stand-in for caput.config (Property / Reader / enum / list_type)."""


class Property(object):
    def __init__(self, default=None, proptype=None, key=None):
        self.proptype = (lambda x: x) if proptype is None else proptype
        self.default = default
        self.key = key
        self.propname = None

    def _name(self, obj):
        if self.propname is None:
            for cls in type(obj).__mro__:
                for k, v in cls.__dict__.items():
                    if v is self:
                        self.propname = "_prop_" + k
                        if self.key is None:
                            self.key = k
        return self.propname

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        name = self._name(obj)
        return obj.__dict__.get(name, self.default)

    def __set__(self, obj, val):
        name = self._name(obj)
        obj.__dict__[name] = None if val is None else self.proptype(val)

    def _from_config(self, obj, cfg):
        self._name(obj)
        if self.key in cfg:
            self.__set__(obj, cfg[self.key])


def enum(options, default=None):
    def _prop(val):
        if val not in options:
            raise ValueError("bad enum value %r" % (val,))
        return val

    return Property(proptype=_prop, default=default)


def list_type(type_=None, length=None, maxlength=None, default=None):
    def _prop(val):
        val = list(val)
        if type_ is not None:
            val = [type_(v) for v in val]
        return val

    return Property(proptype=_prop, default=default)


class Reader(object):
    @classmethod
    def from_config(cls, cfg, *args, **kwargs):
        c = cls(*args, **kwargs)
        c.read_config(cfg)
        return c

    def read_config(self, cfg):
        seen = set()
        for klass in type(self).__mro__:
            for name, prop in klass.__dict__.items():
                if isinstance(prop, Property) and name not in seen:
                    seen.add(name)
                    prop._from_config(self, cfg)
        self._finalise_config()

    def _finalise_config(self):
        pass
