"""TEST INFRASTRUCTURE ONLY.  The order the product's host code imposes on its streams, checked without a device.

The CPU build of the host code (stub_device.cpp) executes every launch at once, in program order, so a missing hipStreamWaitEvent cannot change a result there -- and on the
device it changes one only when the timing is unlucky.  This module turns the stand-in runtime into a race detector instead: it keeps a vector clock per stream and per
recorded event (record = a copy of the stream's clock, wait = a merge, a host synchronisation = a merge into the host's clock, which every later operation starts from), and for
every launch the planes its handler reads (device.view) and writes (device.store), by row window.  An access conflicts with an earlier one of another stream -- write / read,
read / write, write / write on overlapping rows of one plane -- unless the earlier one happens-before it by those clocks; every such pair is a finding.

`drop_wait = k` ignores the k-th hipStreamWaitEvent of the run: the control that shows a missing wait IS found (run.py `order` drops every wait of a steady-state frame in turn and
lists which are needed for the planes the handlers touch).

What it does not see: reads the handlers do not make through device.view (the blue-noise tables, the IBL cubes and their apron copy, host copies), and rows a kernel reads
beyond what the reference's pass reads of a plane (the handlers read whole planes: for a whole-frame chain that is the safe side)."""
import bisect

RT_STREAM_CREATE, RT_STREAM_DESTROY, RT_EVENT_CREATE, RT_EVENT_DESTROY, RT_EVENT_RECORD, RT_STREAM_WAIT, RT_STREAM_SYNC, RT_EVENT_SYNC, RT_WRITE, RT_READ, RT_MALLOC, RT_FREE = range(1, 13)


def _merge(into, other):
    for k, v in other.items():
        if into.get(k, 0) < v:
            into[k] = v


class LaneOrder:
    def __init__(self):
        self.vc = {}        # stream -> {stream: time}
        self.events = {}    # event -> clock at its last record (absent: never recorded = a wait does nothing, as on the device)
        self.host = {}      # what the host has waited for
        self.planes = {}    # base address -> {"w": [(y0, y1, stream, time, name)], "r": [...]}
        self.allocs = []    # sorted [(base, end)]
        self.cur = None     # (stream, clock snapshot, name) of the launch in progress
        self.findings = []  # (kind, plane, rows, earlier launch, earlier stream, later launch, later stream)
        self.waits = 0
        self.drop_wait = None
        self.wait_log = []  # (index, stream, event, was the event recorded) in call order
        self.launches = 0

    # ---- clocks
    def _tick(self, s):
        c = self.vc.setdefault(s, {})
        _merge(c, self.host)  # issued by the host after everything it has waited for
        c[s] = c.get(s, 0) + 1
        return c

    def runtime(self, op, a, b, n):
        a, b = a or 0, b or 0
        if op == RT_STREAM_CREATE:
            self.vc[a] = {}
        elif op == RT_STREAM_DESTROY:
            self.vc.pop(a, None)
        elif op in (RT_EVENT_CREATE, RT_EVENT_DESTROY):
            self.events.pop(a, None)
        elif op == RT_EVENT_RECORD:
            self.events[a] = dict(self._tick(b))
        elif op == RT_STREAM_WAIT:
            k = self.waits
            self.waits += 1
            self.wait_log.append((k, a, b, b in self.events))
            c = self._tick(a)
            if k != self.drop_wait and b in self.events:
                _merge(c, self.events[b])
        elif op == RT_STREAM_SYNC:
            _merge(self.host, self.vc.get(a, {}))
        elif op == RT_EVENT_SYNC:
            _merge(self.host, self.events.get(a, {}))
        elif op in (RT_WRITE, RT_READ):  # memset / memcpy on a stream: an access of the plane that starts there (or of the allocation that holds the address)
            base = self._base_of(a)
            self._access(base, 0, 1 << 30, op == RT_WRITE, b, self._tick(b), "memset/memcpy")
        elif op == RT_MALLOC:
            bisect.insort(self.allocs, (a, a + n))
            self._forget(a, a + n)
        elif op == RT_FREE:
            for c in self.vc.values():  # hipFree waits for the device: everything queued so far is behind the host from here on
                _merge(self.host, c)
            i = bisect.bisect_left(self.allocs, (a, 0))
            if i < len(self.allocs) and self.allocs[i][0] == a:
                self._forget(*self.allocs.pop(i))

    def _forget(self, lo, hi):
        for base in [p for p in self.planes if lo <= p < hi]:
            del self.planes[base]

    def _base_of(self, p):
        return p  # (fills and imports address whole planes by their first byte: the key the launches use)

    # ---- launches
    def begin_launch(self, stream, name):
        self.launches += 1
        self.cur = (stream, dict(self._tick(stream)), name)

    def end_launch(self):
        self.cur = None

    def access(self, img, write):
        if self.cur is None:
            return
        s, clock, name = self.cur
        y0, y1 = (img.y0, img.y0 + img.yn) if (write and img.yn) else (0, img.h)  # a windowed launch writes its window; a handler reads the whole plane
        self._access(img.p, y0, y1, write, s, clock, name)

    def _access(self, base, y0, y1, write, s, clock, name):
        rec = self.planes.setdefault(base, {"w": [], "r": []})
        t = clock.get(s, 0)

        def check(kind, earlier):
            for (a0, a1, es, et, en) in earlier:
                if es != s and a0 < y1 and y0 < a1 and clock.get(es, 0) < et:
                    f = (kind, base, (max(a0, y0), min(a1, y1)), en, es, name, s)
                    if f not in self.findings:
                        self.findings.append(f)

        check("write then write" if write else "write then read", rec["w"])
        if write:
            check("read then write", rec["r"])
            # what this write covers is superseded: a later access ordered behind this write is ordered behind those as well
            rec["w"] = [e for e in rec["w"] if not (y0 <= e[0] and e[1] <= y1)] + [(y0, y1, s, t, name)]
            rec["r"] = [e for e in rec["r"] if not (y0 <= e[0] and e[1] <= y1)]
        else:
            rec["r"] = [e for e in rec["r"] if not (e[2] == s and e[0] == y0 and e[1] == y1)] + [(y0, y1, s, t, name)]  # (a stream is in order: its latest read of the rows stands for all)

    # ---- reporting
    def describe(self, streams=None):
        names = streams or {}
        out = []
        for kind, base, rows, en, es, ln, ls in self.findings:
            out.append(f"{kind}: {en} (stream {names.get(es, hex(es))}) and {ln} (stream {names.get(ls, hex(ls))}) on plane {base:#x} rows {rows[0]}..{rows[1]} without an order")
        return out
