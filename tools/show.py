import json,sys
for l in sys.stdin:
    r=json.loads(l)
    if "error" in r: print(r); continue
    print({k:r[k] for k in r if k in ("effort","W","E","S","D","call_us","mul_us","call_us_by_streams")})
