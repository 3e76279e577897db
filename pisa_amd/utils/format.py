"""Small text helpers with the reference's names (pisa/utils/format.py: `split` :169-245, `arg_to_tuple` :277-300,
`hrbool2bool` :554-576, `text2tex` :656-680, `make_valid_python_name` :815-825, `timediff`, `timestamp`)."""
import re
import time
from collections.abc import Iterable, Sequence

__all__ = ["split", "arg_to_tuple", "arg_str_seq_none", "hrbool2bool", "text2tex", "tex_dollars", "is_tex", "make_valid_python_name",
           "timediff", "timestamp"]

_WHITESPACE = re.compile(r"\s")


def split(string, sep=",", force_case=None, parse_func=None):
    """the parts of `string` between `sep`, stripped; a sequence of strings is split element by element; optionally
    lower- / upper-cased and converted"""
    funcs = []
    if force_case == "lower":
        funcs.append(str.lower)
    elif force_case == "upper":
        funcs.append(str.upper)
    elif force_case is not None:
        raise ValueError("`force_case` must be one of (None, 'upper', 'lower'); got %r" % (force_case,))
    if parse_func is not None:
        funcs.append(parse_func)
    if string is None:
        return []
    if isinstance(string, str):
        parts = [x.strip() for x in str.split(string, sep)]
    elif isinstance(string, Iterable):
        parts = [p for x in string for p in split(x, sep)]
    else:
        raise TypeError("Unhandled type %s" % type(string))
    for f in funcs:
        parts = [f(p) for p in parts]
    return parts


def arg_to_tuple(arg):
    """None -> (), a string or a non-sequence -> (arg,), a sequence -> tuple(arg)"""
    if arg is None:
        return ()
    if isinstance(arg, str) or not isinstance(arg, (Sequence, Iterable)):
        return (arg,)
    return tuple(arg)


def arg_str_seq_none(inputs, name):
    """a comma-separated string or a sequence of strings as a lower-case list without whitespace; None stays None"""
    if inputs is None:
        return None
    if isinstance(inputs, str):
        inputs = [inputs]
    if not isinstance(inputs, (Iterable, Sequence)):
        raise TypeError("Argument `%s` must be one of string, sequence of strings or None; got %s" % (name, type(inputs)))
    out = []
    for x in inputs:
        if not isinstance(x, str):
            raise TypeError("Argument `%s` contains a non-string: %r" % (name, x))
        out.extend(_WHITESPACE.sub("", p).lower() for p in x.split(","))
    return out


def hrbool2bool(s):
    """'true' / 't' / 'yes' / 'y' / '1' / 'on' and their opposites, any case"""
    s = str(s).strip().lower()
    if s in ("true", "t", "yes", "y", "1", "on"):
        return True
    if s in ("false", "f", "no", "n", "0", "off"):
        return False
    raise ValueError('Could not parse input "%s" to bool.' % s)


def text2tex(txt):
    """plain text with TeX's special characters escaped and spaces kept"""
    if txt is None:
        return ""
    out = str(txt).replace("\\", r"\backslash ")
    for ch in "%$#_{}&":
        out = out.replace(ch, "\\" + ch)
    return out.replace("^", r"\^{}").replace("~", r"\sim ").replace(" ", r"\;")


def tex_dollars(s):
    stripped = s.strip()
    return stripped if stripped.startswith("$") and stripped.endswith("$") else "$%s$" % stripped


def is_tex(s):
    if s is None:
        return False
    s = str(s)
    return any(tok in s for tok in ("\\", "$", "_{", "^{"))


def make_valid_python_name(name):
    """characters outside [0-9a-zA-Z_] become '_', anything before the first letter / underscore is dropped"""
    name = re.sub(r"[^0-9a-zA-Z_]", "_", str(name))
    return re.sub(r"^[^a-zA-Z_]+", "", name)


def timediff(dt_sec, hms_always=False, sec_decimals=3):
    """seconds as '1 d 02:03:04.567'-style text"""
    sign = "-" if dt_sec < 0 else ""
    dt = abs(float(dt_sec))
    days, rem = divmod(dt, 86400)
    h, rem = divmod(rem, 3600)
    m, s = divmod(rem, 60)
    if not hms_always and days == 0 and h == 0 and m == 0:
        return "%s%.*f sec" % (sign, sec_decimals, s)
    body = "%02d:%02d:%0*.*f" % (h, m, 3 + sec_decimals if sec_decimals else 2, sec_decimals, s)
    return "%s%s%s" % (sign, "%d d " % days if days else "", body)


def timestamp(d=True, t=True, tz=True, utc=False, winsafe=False):
    """ISO-8601-like 'YYYY-MM-DDTHH:MM:SS+ZZZZ' of now"""
    now = time.gmtime() if utc else time.localtime()
    parts = []
    if d:
        parts.append(time.strftime("%Y-%m-%d", now))
    if t:
        parts.append(time.strftime("%H%M%S" if winsafe else "%H:%M:%S", now) + (("Z" if utc else time.strftime("%z", now)) if tz else ""))
    return "T".join(parts)
