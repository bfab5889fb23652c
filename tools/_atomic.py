"""Atomic JSON writer for the profile summarisers: the document is serialised to a string FIRST
(a value json cannot encode raises before any byte is on disk), written to a temporary file beside
the target and moved into place with os.replace.  Round 4 lost four PMC summaries to a plain
`open(path, "w")` + `json.dump` that died after the opening brace."""
import json
import os


def write_json(path, doc, **kw):
  text = json.dumps(doc, allow_nan=False, **kw)       # (NaN / Infinity are not JSON: fail here)
  json.loads(text)                      # what we are about to commit parses
  tmp = "%s.tmp.%d" % (path, os.getpid())
  with open(tmp, "w") as f:
    f.write(text)
    f.write("\n")
    f.flush()
    os.fsync(f.fileno())
  os.replace(tmp, path)
