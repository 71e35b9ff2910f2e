"""Print dotted fields of the last JSON line of a file:  jget.py file a.b c ..."""
import json
import sys

try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
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
