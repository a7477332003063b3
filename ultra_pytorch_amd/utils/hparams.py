"""HParams — the "name=value,name=[v1,v2]" hyper-parameter strings of the settings JSON.

An independent, compact implementation of the grammar the reference's ultra/utils/hparams.py accepts
(PARAM_RE :17-24, parse_values :160-259, HParams.parse :418-438): values are typed by the DEFAULT's type,
bools accept true/false/1/0, lists are comma separated inside [], `name[i]=v` sets one list element, unknown
names are ignored with a printed notice, and re-assigning a name inside one string is an error.
"""
import re

_CLAUSE = re.compile(r"\s*(?P<name>[a-zA-Z]\w*)\s*(\[\s*(?P<index>\d+)\s*\])?\s*=\s*"
                     r"((?P<val>[^,\[]*)|\[(?P<vals>[^\]]*)\])\s*($|,)")


def _to_bool(s):
    s = s.strip()
    if s in ("true", "True"):
        return True
    if s in ("false", "False"):
        return False
    return bool(int(s))


class HParams(object):
    def __init__(self, **defaults):
        object.__setattr__(self, "_types", {})
        for k, v in defaults.items():
            self.add_hparam(k, v)

    def add_hparam(self, name, value):
        if name in self._types:
            raise ValueError("Hyperparameter name is reserved or duplicated: %s" % name)
        if isinstance(value, (list, tuple)):
            if not value:
                raise ValueError("Multi-valued hyperparameters cannot be empty: %s" % name)
            self._types[name] = (type(value[0]), True)
            value = list(value)
        else:
            self._types[name] = (type(value), False)
        object.__setattr__(self, name, value)

    def _cast(self, name, typ, text):
        try:
            if typ is bool:
                return _to_bool(text)
            if typ is int:
                f = float(text)
                if f != int(f):
                    raise ValueError
                return int(f)
            if typ is str:
                return text.strip()
            return typ(text)
        except (ValueError, TypeError):
            raise ValueError("Could not parse hparam '%s' of type '%s' with value '%s'" % (name, typ.__name__, text))

    def parse(self, values, ignore_unknown_hyperparameters=True):
        seen, pos = set(), 0
        values = values or ""
        while pos < len(values):
            m = _CLAUSE.match(values, pos)
            if not m:
                raise ValueError("Malformed hyperparameter value: %s" % values[pos:])
            pos = m.end()
            name, index, val, vals = m.group("name"), m.group("index"), m.group("val"), m.group("vals")
            if name not in self._types:
                if not ignore_unknown_hyperparameters:
                    raise ValueError("Unknown hyperparameter type for %s" % name)
                print("Unknown hyperparameter type for %s" % name)
                continue
            typ, is_list = self._types[name]
            key = (name, index)
            if key in seen or (name, None) in seen and index is None:
                raise ValueError("Multiple assignments to variable '%s' in %s" % (name, values))
            seen.add(key)
            if vals is not None:
                if index is not None:
                    raise ValueError("Assignment of a list to a list index: %s" % name)
                if not is_list:
                    raise ValueError("Must not pass a list for single-valued parameter: %s" % name)
                items = [v for v in (x.strip() for x in vals.split(",")) if v != ""]
                object.__setattr__(self, name, [self._cast(name, typ, v) for v in items])
            elif index is not None:
                if not is_list:
                    raise ValueError("Index on a single-valued parameter: %s" % name)
                lst = list(getattr(self, name))
                i = int(index)
                while len(lst) <= i:
                    lst.append(lst[-1])
                lst[i] = self._cast(name, typ, val)
                object.__setattr__(self, name, lst)
            else:
                if is_list:
                    raise ValueError("Must pass a list for multi-valued parameter: %s." % name)
                object.__setattr__(self, name, self._cast(name, typ, val))
        return self

    def values(self):
        return {k: getattr(self, k) for k in self._types}

    def __contains__(self, key):
        return key in self._types

    def __repr__(self):
        return "HParams(%s)" % ", ".join("%s=%r" % kv for kv in sorted(self.values().items()))
