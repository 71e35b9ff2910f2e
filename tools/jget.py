"""Print dotted fields of the last JSON line of a file (or of an indented JSON file):  jget.py file a.b c ..."""
import json
import sys

try:
    text = open(sys.argv[1]).read().strip()
    try:
        d = json.loads(text.splitlines()[-1])  # a bench line: one JSON object on the last line
    except ValueError:
        d = json.loads(text)                   # an indented JSON file
except Exception as e:  # noqa: BLE001
    print("no JSON (%s)" % e)
    sys.exit(0)
res = []
for key in sys.argv[2:]:
    v = d
    for part in key.split("."):
        v = v.get(part) if isinstance(v, dict) else None
    res.append("%s=%s" % (key, v))
print(" ".join(res))
