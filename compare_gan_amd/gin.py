"""A small dependency-free re-implementation of the part of gin-config the reference uses.

The reference wires everything through gin-config 0.1.4 (setup.py:33): `@gin.configurable`
decorators with explicit names and white/blacklists, `gin.REQUIRED`, `@reference` values and
module-qualified selectors.  gin is not installed in this environment, and the
drop-in contract is that `example_configs/*.gin` parse and bind unchanged, so this module provides
the same surface: configurable / external_configurable / REQUIRED / bind_parameter /
query_parameter / parse_config / parse_config_file(s_and_bindings) / clear_config /
operative_config_str / config_str.

Supported syntax: `#` comments, `selector.arg = <python literal>`, `@configurable` references
(and `@configurable()` evaluated references), `%macro` references with `macro = value`
definitions, line continuation inside brackets, `scope/selector.arg` (scope is recorded and
ignored, as no example config uses scopes).
"""
import ast
import functools
import inspect
import re

REQUIRED = type("Required", (), {"__repr__": lambda self: "gin.REQUIRED"})()

_REGISTRY = {}      # full name "module.path.Name" -> _Configurable
_BINDINGS = {}      # full name -> {arg: value}
_UNRESOLVED = {}    # selector text -> {arg: value} for selectors not registered yet
_MACROS = {}
_OPERATIVE = {}     # full name -> {arg: value} actually injected


class _Ref(object):
  """`@name` (evaluate=False) or `@name()` (evaluate=True) value."""

  def __init__(self, selector, evaluate):
    self.selector, self.evaluate = selector, evaluate

  def resolve(self):
    c = _lookup(self.selector)
    return c.wrapped() if self.evaluate else c.wrapped

  def __repr__(self):
    return "@" + self.selector + ("()" if self.evaluate else "")


class _Macro(object):
  def __init__(self, name):
    self.name = name

  def resolve(self):
    if self.name not in _MACROS:
      raise ValueError("gin: undefined macro %%%s" % self.name)
    return _resolve(_MACROS[self.name])

  def __repr__(self):
    return "%" + self.name


class _Configurable(object):
  def __init__(self, name, module, fn, wrapped, whitelist, blacklist):
    self.name, self.module = name, module
    self.full = (module + "." + name) if module else name
    self.fn, self.wrapped = fn, wrapped
    self.whitelist, self.blacklist = whitelist, blacklist

  def allows(self, arg):
    if self.whitelist is not None:
      return arg in self.whitelist
    if self.blacklist is not None:
      return arg not in self.blacklist
    return True


def _resolve(v):
  if isinstance(v, (_Ref, _Macro)):
    return v.resolve()
  if isinstance(v, list):
    return [_resolve(x) for x in v]
  if isinstance(v, tuple):
    return tuple(_resolve(x) for x in v)
  if isinstance(v, dict):
    return {k: _resolve(x) for k, x in v.items()}
  return v


def _matches(full, selector):
  return full == selector or full.endswith("." + selector)


def _lookup(selector):
  hits = [c for f, c in _REGISTRY.items() if _matches(f, selector)]
  if not hits:
    raise ValueError("gin: no configurable matching '%s'" % selector)
  if len(hits) > 1:
    exact = [c for c in hits if c.full == selector]
    if len(exact) == 1:
      return exact[0]
    raise ValueError("gin: ambiguous selector '%s' matches %s" % (
        selector, sorted(c.full for c in hits)))
  return hits[0]


def _bind(selector, arg, value):
  hits = [c for f, c in _REGISTRY.items() if _matches(f, selector)]
  if len(hits) > 1:
    exact = [c for c in hits if c.full == selector]
    hits = exact if len(exact) == 1 else hits
  if len(hits) > 1:
    raise ValueError("gin: ambiguous selector '%s'" % selector)
  if not hits:
    _UNRESOLVED.setdefault(selector, {})[arg] = value
    return
  c = hits[0]
  if not c.allows(arg):
    raise ValueError("gin: '%s' is not a configurable parameter of '%s'" % (arg, c.full))
  params = _signature_params(c.fn)
  if arg not in params and not _has_varkw(c.fn):
    raise ValueError("gin: configurable '%s' has no parameter '%s'" % (c.full, arg))
  _BINDINGS.setdefault(c.full, {})[arg] = value


def _signature_params(fn):
  target = fn.__init__ if inspect.isclass(fn) else fn
  try:
    return [p for p in inspect.signature(target).parameters if p != "self"]
  except (TypeError, ValueError):
    return []


def _has_varkw(fn):
  target = fn.__init__ if inspect.isclass(fn) else fn
  try:
    return any(p.kind == p.VAR_KEYWORD for p in inspect.signature(target).parameters.values())
  except (TypeError, ValueError):
    return True


def _make_wrapper(fn, holder):
  sig = inspect.signature(fn)
  has_self = holder.get("is_init", False)

  @functools.wraps(fn)
  def wrapper(*args, **kwargs):
    c = holder["c"]
    bindings = _BINDINGS.get(c.full, {})
    try:
      bound = sig.bind_partial(*args, **kwargs)
      given = set(bound.arguments)
    except TypeError:
      given = set(kwargs)
    inject = {}
    for arg, value in bindings.items():
      if arg in given:
        continue
      inject[arg] = _resolve(value)
    if inject:
      _OPERATIVE.setdefault(c.full, {}).update({k: bindings[k] for k in inject})
    kwargs = dict(kwargs)
    kwargs.update(inject)
    # gin.REQUIRED defaults must have been supplied by now
    for pname, p in sig.parameters.items():
      if p.default is REQUIRED and pname not in kwargs and pname not in given:
        raise ValueError("gin: required parameter '%s' of '%s' was not bound" % (pname, c.full))
    return fn(*args, **kwargs)

  wrapper.__gin_wrapped__ = fn
  del has_self
  return wrapper


def _register(fn, name, module, whitelist, blacklist):
  if whitelist is not None and blacklist is not None:
    raise ValueError("gin: specify at most one of whitelist / blacklist")
  name = name or fn.__name__
  if module is None:
    if "." in name:
      module, name = name.rsplit(".", 1)
    else:
      module = getattr(fn, "__module__", None)
  holder = {}
  if inspect.isclass(fn):
    holder["is_init"] = True
    orig_init = fn.__dict__.get("__init__")
    if orig_init is None:
      # class without its own __init__: give it one so that the wrapper is per-class
      parent_init = fn.__init__

      def orig_init(self, *a, **k):  # pylint: disable=function-redefined
        parent_init(self, *a, **k)
      orig_init.__signature__ = inspect.signature(parent_init)
    wrapped_init = _make_wrapper(orig_init, holder)
    fn.__init__ = wrapped_init
    wrapped = fn
  else:
    wrapped = _make_wrapper(fn, holder)
  c = _Configurable(name, module, fn, wrapped, whitelist, blacklist)
  holder["c"] = c
  if c.full in _REGISTRY and _REGISTRY[c.full].fn is not fn:
    raise ValueError("gin: configurable '%s' registered twice" % c.full)
  _REGISTRY[c.full] = c
  # bindings parsed before this configurable was imported
  for sel in [s for s in _UNRESOLVED if _matches(c.full, s)]:
    for arg, value in _UNRESOLVED.pop(sel).items():
      _bind(c.full, arg, value)
  return wrapped


def configurable(name_or_fn=None, module=None, whitelist=None, blacklist=None):
  """@gin.configurable, @gin.configurable("name"), @gin.configurable(whitelist=[...])."""
  if callable(name_or_fn):
    return _register(name_or_fn, None, module, whitelist, blacklist)

  def deco(fn):
    return _register(fn, name_or_fn, module, whitelist, blacklist)
  return deco


def external_configurable(fn, name=None, module=None, whitelist=None, blacklist=None):
  """Registers a function that cannot be decorated; returns the configurable version."""
  if inspect.isclass(fn):
    fn = type(fn.__name__, (fn,), {"__module__": fn.__module__})
  return _register(fn, name, module, whitelist, blacklist)


def bind_parameter(binding_key, value):
  selector, arg = binding_key.rsplit(".", 1)
  if "/" in selector:
    selector = selector.rsplit("/", 1)[1]
  _bind(selector, arg, value)


def query_parameter(binding_key):
  selector, arg = binding_key.rsplit(".", 1)
  c = _lookup(selector)
  b = _BINDINGS.get(c.full, {})
  if arg not in b:
    raise ValueError("gin: no binding for '%s'" % binding_key)
  return b[arg]


def clear_config(clear_constants=False):
  del clear_constants
  _BINDINGS.clear()
  _UNRESOLVED.clear()
  _MACROS.clear()
  _OPERATIVE.clear()


# ------------------------------------------------------------------------------------------------
# parsing
# ------------------------------------------------------------------------------------------------
_REF_RE = re.compile(r"@([A-Za-z_][\w./]*)(\(\))?")
_MACRO_RE = re.compile(r"%([A-Za-z_][\w.]*)")


def _parse_value(text):
  text = text.strip()
  refs = []

  def ref_sub(m):
    refs.append(_Ref(m.group(1), bool(m.group(2))))
    return "__gin_ref_%d__" % (len(refs) - 1)

  def macro_sub(m):
    refs.append(_Macro(m.group(1)))
    return "__gin_ref_%d__" % (len(refs) - 1)

  # protect string literals from @ / % substitution
  parts = re.split(r"""("(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*')""", text)
  for i in range(0, len(parts), 2):
    parts[i] = _MACRO_RE.sub(macro_sub, _REF_RE.sub(ref_sub, parts[i]))
  text2 = "".join(parts)
  tree = ast.parse(text2, mode="eval")

  def conv(node):
    if isinstance(node, ast.Name):
      m = re.match(r"__gin_ref_(\d+)__$", node.id)
      if m:
        return refs[int(m.group(1))]
      if node.id in ("True", "False", "None"):
        return {"True": True, "False": False, "None": None}[node.id]
      raise ValueError("gin: cannot parse value %r" % text)
    if isinstance(node, (ast.List, ast.Tuple)):
      vals = [conv(e) for e in node.elts]
      return vals if isinstance(node, ast.List) else tuple(vals)
    if isinstance(node, ast.Dict):
      return {conv(k): conv(v) for k, v in zip(node.keys, node.values)}
    return ast.literal_eval(node)

  return conv(tree.body)


def _logical_lines(text):
  buf, depth = "", 0
  for raw in text.splitlines():
    line = re.split(r"""#(?=(?:[^"']|"[^"]*"|'[^']*')*$)""", raw, 1)[0].rstrip()
    if not line.strip() and depth == 0:
      continue
    buf += (" " if buf else "") + line.strip()
    depth = buf.count("(") + buf.count("[") + buf.count("{") - buf.count(")") - buf.count(
        "]") - buf.count("}")
    if buf.endswith("\\"):
      buf = buf[:-1]
      continue
    if depth <= 0:
      yield buf
      buf, depth = "", 0
  if buf.strip():
    yield buf


def parse_config(bindings, skip_unknown=False):
  """Parses a config string (or list of binding strings)."""
  if isinstance(bindings, (list, tuple)):
    bindings = "\n".join(bindings)
  for line in _logical_lines(bindings):
    if line.startswith("import ") or line.startswith("include "):
      continue  # modules are imported by the package itself
    if "=" not in line:
      raise ValueError("gin: cannot parse line %r" % line)
    key, value = line.split("=", 1)
    key = key.strip()
    value = _parse_value(value)
    if "." not in key:
      _MACROS[key] = value
      continue
    selector, arg = key.rsplit(".", 1)
    if "/" in selector:
      selector = selector.rsplit("/", 1)[1]
    try:
      _bind(selector, arg, value)
    except ValueError:
      if not skip_unknown:
        raise


def parse_config_file(path, skip_unknown=False):
  with open(path) as f:
    parse_config(f.read(), skip_unknown=skip_unknown)


def parse_config_files_and_bindings(config_files, bindings, finalize_config=True,
                                    skip_unknown=False):
  del finalize_config
  for path in config_files or []:
    parse_config_file(path, skip_unknown=skip_unknown)
  if bindings:
    parse_config(bindings, skip_unknown=skip_unknown)


def unresolved_selectors():
  """Selectors bound in a config that no imported configurable matches (finalize-time check)."""
  return sorted(_UNRESOLVED)


def _fmt(v):
  return repr(v)


def _dump(table):
  out = []
  for full in sorted(table):
    c = _REGISTRY.get(full)
    name = c.name if c and sum(1 for k in _REGISTRY.values() if k.name == c.name) == 1 else full
    for arg in sorted(table[full]):
      out.append("%s.%s = %s" % (name, arg, _fmt(table[full][arg])))
    out.append("")
  return "\n".join(out)


def config_str():
  return _dump(_BINDINGS)


def operative_config_str():
  """Bindings that were actually injected into a call so far (runner_lib.py:319)."""
  return _dump(_OPERATIVE)
