"""Offline census of a saved phase trace (tools/trace_group.py --save): per microsecond of the launch, how many workgroups
have loads outstanding, sit between the landed tile and their first store (reduce / meet / parameters), are storing
(first store issued .. stores acknowledged), and how many of the resident slots are empty."""
import sys
import numpy as np
for path in sys.argv[1:]:
    d = np.load(path)
    t = d['t'].astype(np.float64) / 100.0
    t0 = t[:, 0].min()
    T = t - t0
    end = T[:, 8]
    span = end.max()
    nb = int(span) + 2
    ph = np.zeros((4, nb))
    for w in range(T.shape[0]):
        s, red, par, si, e = T[w, 0], T[w, 2], T[w, 5], T[w, 6], end[w]
        ph[0, int(s):int(red) + 1] += 1        # loads in flight
        ph[1, int(red) + 1:int(par) + 1] += 1  # dead (reduce, meet, parameters)
        ph[2, int(par) + 1:int(e) + 1] += 1    # Q/DQ + stores until drained
    tot = ph[:3].sum(0)
    slots = tot.max()
    print('%s: %d workgroups, span %.1f us, plain launch %.1f us, peak residents %d' % (path, T.shape[0], span, float(d['plain_us']), slots))
    mid = slice(int(nb * 0.15), int(nb * 0.85))
    print('   steady state (15..85 %% of the launch): mean residents %.0f | loading %.0f | between landed and first store %.0f | storing %.0f' % (
        tot[mid].mean(), ph[0, mid].mean(), ph[1, mid].mean(), ph[2, mid].mean()))
    # how phased is the chip?  fraction of microseconds in which > 70 % of the residents are in one phase
    frac_l = (ph[0, mid] > 0.7 * tot[mid]).mean(); frac_s = (ph[2, mid] > 0.7 * tot[mid]).mean()
    print('   microseconds with > 70 %% of residents loading: %.2f, storing: %.2f' % (frac_l, frac_s))
    life = end - T[:, 0]
    print('   lifetime p50 %.1f us; per phase p50: issue %.1f, land+reduce %.1f, publish+meet+pairs+params %.1f, qdq+store issue %.1f, drain %.1f' % (
        np.median(life), np.median(T[:, 1] - T[:, 0]), np.median(T[:, 2] - T[:, 1]), np.median(T[:, 5] - T[:, 2]), np.median(T[:, 6] - T[:, 5]), np.median(end - T[:, 6])))
    step = max(1, nb // 60)
    print('   t: load/dead/store  ' + ' '.join('%d:%d/%d/%d' % (i, ph[0, i], ph[1, i], ph[2, i]) for i in range(0, nb, step)))
