"""Plugin calling convention and parameter overview (reference: compare_gan/utils.py:92-174)."""
import inspect


def _has_arg(fn, arg_name):
  """True if `fn` accepts a keyword argument `arg_name` (looks through gin / functools wrappers)."""
  while hasattr(fn, "__gin_wrapped__") or hasattr(fn, "__wrapped__"):
    fn = getattr(fn, "__gin_wrapped__", None) or fn.__wrapped__
  spec = inspect.getfullargspec(fn)
  if spec.varkw:
    return True
  return arg_name in spec.args or arg_name in spec.kwonlyargs


def call_with_accepted_args(fn, **kwargs):
  """Calls `fn` only with the keyword arguments that `fn` accepts (utils.py:92-96)."""
  kwargs = {k: v for k, v in kwargs.items() if _has_arg(fn, k)}
  return fn(**kwargs)


def get_parameter_overview(variables, limit=40):
  """Table with name, shape and size of (name, tensor) pairs plus the total (utils.py:99-174)."""
  rows = [(n, tuple(v.shape), v.numel()) for n, v in variables]
  total = sum(r[2] for r in rows)
  shown = rows if limit is None else rows[:limit]
  w = max([len(r[0]) for r in shown] + [4])
  lines = ["%-*s  %-22s %12s" % (w, "Name", "Shape", "Size")]
  lines += ["%-*s  %-22s %12s" % (w, n, str(s), "{:,}".format(c)) for n, s, c in shown]
  if limit is not None and len(rows) > limit:
    lines.append("[...and %d more variables.]" % (len(rows) - limit))
  lines.append("Total: {:,}".format(total))
  return "\n".join(lines)
