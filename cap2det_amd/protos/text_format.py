"""proto2 text-format reader/writer for the schema in `schema.py`.

Replaces `google.protobuf.text_format.Merge` as used by the reference
(train/trainer_main.py:27-33, models/label_extractor_test.py:24-31): same call shape
`text_format.Merge(text, message)`, `ParseError` on malformed input.  Supports the syntax
found in `configs/*.pbtxt`: nested `name { }` / `name < >` blocks, `name: value`, the
extension form `[Cap2DetModel.ext] { }`, single/double quoted strings with escapes and
adjacent-literal concatenation, repeated scalars (repeated keys or `[a, b]` lists),
enums by identifier or number, `#` comments, optional `,`/`;` separators.
"""
import re

from cap2det_amd.protos.message import Message, unwrap


class ParseError(ValueError):
  pass


_TOKEN = re.compile(r"""
    (?P<ws>\s+|\#[^\n]*) |
    (?P<str>"(?:\\.|[^"\\\n])*"|'(?:\\.|[^'\\\n])*') |
    (?P<ext>\[\s*[A-Za-z_][\w.]*\s*\]) |
    (?P<num>[-+]?(?:(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?[fF]?|0[xX][0-9a-fA-F]+|inf|nan|infinity)\b) |
    (?P<id>[A-Za-z_][\w.]*) |
    (?P<sym>[{}<>:,;\[\]])
""", re.VERBOSE)

_ESC = {"n": "\n", "t": "\t", "r": "\r", "\\": "\\", "'": "'", '"': '"', "0": "\0",
        "a": "\a", "b": "\b", "f": "\f", "v": "\v", "?": "?"}


def _unescape(body):
  out, i = [], 0
  while i < len(body):
    c = body[i]
    if c == "\\" and i + 1 < len(body):
      n = body[i + 1]
      if n in _ESC:
        out.append(_ESC[n]); i += 2
      elif n in "xX":
        m = re.match(r"[0-9a-fA-F]{1,2}", body[i + 2:])
        if not m:
          raise ParseError("bad \\x escape")
        out.append(chr(int(m.group(0), 16))); i += 2 + len(m.group(0))
      elif n.isdigit():
        m = re.match(r"[0-7]{1,3}", body[i + 1:])
        out.append(chr(int(m.group(0), 8))); i += 1 + len(m.group(0))
      else:
        raise ParseError("bad escape \\%s" % n)
    else:
      out.append(c); i += 1
  return "".join(out)


def _tokenize(text):
  pos, toks, line = 0, [], 1
  while pos < len(text):
    m = _TOKEN.match(text, pos)
    if not m:
      raise ParseError("%d: unexpected character %r" % (line, text[pos]))
    kind = m.lastgroup
    val = m.group(kind)
    if kind != "ws":
      toks.append((kind, val, line))
    line += val.count("\n")
    pos = m.end()
  toks.append(("eof", "", line))
  return toks


class _Parser(object):
  def __init__(self, text):
    self.toks = _tokenize(text)
    self.i = 0

  def peek(self):
    return self.toks[self.i]

  def next(self):
    t = self.toks[self.i]
    self.i += 1
    return t

  def error(self, msg):
    raise ParseError("%d: %s" % (self.peek()[2], msg))

  def accept(self, sym):
    k, v, _ = self.peek()
    if k == "sym" and v == sym:
      self.i += 1
      return True
    return False

  def parse_message(self, msg, closer):
    msg = unwrap(msg)
    cls = type(msg)
    while True:
      k, v, _ = self.peek()
      if k == "eof":
        if closer is not None:
          self.error("unexpected end of input, expected '%s'" % closer)
        return
      if k == "sym" and closer is not None and v == closer:
        self.i += 1
        return
      if k == "sym" and v in ",;":
        self.i += 1
        continue
      if k == "ext":
        self.i += 1
        name = v[1:-1].strip()
        fd = cls._extensions.get(name)
        if fd is None:
          self.error('Extension "%s" not registered for %s' % (name, cls._name))
      elif k == "id":
        self.i += 1
        fd = cls._fields.get(v)
        if fd is None:
          self.error('Message type "%s" has no field named "%s"' % (cls._name, v))
      else:
        self.error("expected field name, got %r" % v)
      self.parse_field(msg, fd)

  def parse_field(self, msg, fd):
    if fd.is_message:
      self.accept(":")
      if self.accept("{"):
        closer = "}"
      elif self.accept("<"):
        closer = ">"
      else:
        self.error("expected '{' for message field %s" % fd.full_name)
      if fd.label == "repeated":
        child = msg._get(fd).add()
      else:
        key = msg._key(fd)
        child = msg._values.get(key)
        if child is None:
          child = fd.message_class()()
          msg._set_message(fd, child)
      self.parse_message(child, closer)
      return
    if not self.accept(":"):
      self.error("expected ':' after scalar field %s" % fd.full_name)
    if fd.label == "repeated" and self.accept("["):
      if not self.accept("]"):
        while True:
          msg._get(fd).append(self.parse_scalar(fd))
          if self.accept("]"):
            break
          if not self.accept(","):
            self.error("expected ',' or ']' in list")
      return
    val = self.parse_scalar(fd)
    if fd.label == "repeated":
      msg._get(fd).append(val)
    else:
      msg._clear_oneof_siblings(fd)
      msg._values[msg._key(fd)] = val

  def parse_scalar(self, fd):
    k, v, _ = self.next()
    t = fd.type
    try:
      if t == "string":
        if k != "str":
          raise ValueError("expected string")
        parts = [_unescape(v[1:-1])]
        while self.peek()[0] == "str":
          parts.append(_unescape(self.next()[1][1:-1]))
        return "".join(parts)
      if t in ("int32", "int64"):
        if k != "num":
          raise ValueError("expected integer")
        return int(v, 0)
      if t in ("float", "double"):
        if k == "num":
          return float(v.rstrip("fF")) if not v.lower().startswith("0x") else float(int(v, 16))
        if k == "id" and v.lower() in ("inf", "infinity", "nan"):
          return float(v)
        raise ValueError("expected number")
      if t == "bool":
        if (k == "id" and v in ("true", "True", "t")) or (k == "num" and v == "1"):
          return True
        if (k == "id" and v in ("false", "False", "f")) or (k == "num" and v == "0"):
          return False
        raise ValueError("expected bool")
      if t == "enum":
        if k == "id":
          if v not in fd.enum_values:
            raise ValueError("unknown enum value %s" % v)
          return fd.enum_values[v]
        if k == "num" and int(v, 0) in fd.enum_values.values():
          return int(v, 0)
        raise ValueError("expected enum")
    except ValueError as e:
      raise ParseError("%d: %s for field %s (got %r)" % (self.toks[self.i - 1][2], e,
                                                        fd.full_name, v))
    raise ParseError("unsupported scalar type %s" % t)


def Merge(text, message):
  """Parses `text` into `message` (merging), returns `message`."""
  if isinstance(text, bytes):
    text = text.decode("utf-8")
  if not isinstance(unwrap(message), Message):
    raise TypeError("message must be a schema Message")
  _Parser(text).parse_message(message, None)
  return message


Parse = Merge


def _fmt_scalar(fd, v):
  if fd.type == "string":
    return '"%s"' % v.replace("\\", "\\\\").replace('"', '\\"').replace("\n", "\\n")
  if fd.type == "bool":
    return "true" if v else "false"
  if fd.type == "enum":
    for k, n in fd.enum_values.items():
      if n == v:
        return k
  if fd.type in ("float", "double"):
    return repr(float(v))
  return str(v)


def MessageToString(message, indent=0):
  message = unwrap(message)
  pad = " " * indent
  lines = []
  for fd, v in message.ListFields():
    name = "[%s]" % fd.full_name if fd.is_extension else fd.name
    items = v if fd.label == "repeated" else [v]
    for item in items:
      if fd.is_message:
        lines.append("%s%s {\n%s%s}\n" % (pad, name, MessageToString(item, indent + 2), pad))
      else:
        lines.append("%s%s: %s\n" % (pad, name, _fmt_scalar(fd, item)))
  return "".join(lines)
