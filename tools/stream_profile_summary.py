import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')][1:]
for d in rows:
    t = d['top'][0]
    print(sys.argv[1], 'total', round(d['total_ms']), 'kernel', round(d['fit_kernel_ms']), 'sum_target_s', round(d['sum_target_s'], 1), 'top: dur', round(t['dur_ms']), 'fold', round(t.get('fold_ms', 0)), 'upd', round(t.get('upd_ms', 0)), 'gap', round(t.get('gap_ms', 0)), 'consumer_wait(spec only)', round(t.get('wait_ms', 0)), 'folded', t.get('folded_entries'))
