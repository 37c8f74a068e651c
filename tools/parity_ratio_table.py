"""gpurun_out/parity_ratios.jsonl (written by tests/test_gpu_parity.py on the GPU box) -> the table committed under profiles/:
for every (test, fixture, recorded step) the HIP path's error against the reference's recorded fp32 output, the frozen
conditioning floor of that state (tests/golden/conditioning_floor.json) and their ratio.  argv: jsonl [commit]"""
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
commit = sys.argv[2] if len(sys.argv) > 2 else '?'
print(f'# (commit {commit}): HIP error / conditioning floor for every checked sampler step\n')
print('`python -m pytest tests -m gpu` on 1x MI355X.  err = max-abs error of the HIP output against the REFERENCE\'s recorded fp32 output, relative to '
      'max|reference|; floor = largest distance of a 12-member fp32 ensemble of the reference\'s own dataflow (rows permuted, coordinates '
      '+-1 ulp) from its float64 evaluation on the same state (`oracle/make_conditioning_floor.py`, frozen in '
      '`tests/golden/conditioning_floor.json`); the tests assert err <= max(5 x 2e-5, 3 x floor) with both constants fixed in '
      '`tests/helpers.py`.  Columns: v (atom-type logits), x0 (coordinates), bond (bond-type logits).\n')
print('| test | fixture | step | err v | err x0 | err bond | floor v | floor x0 | floor bond | ratio v | ratio x0 | ratio bond |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|')
worst, over = 0.0, 0
for r in rows:
    ratio = [e / f for e, f in zip(r['err'], r['floor'])]
    worst = max(worst, max(ratio))
    over += sum(1 for x in ratio if x > 3.0)
    print('| %s | %s | %d | %s | %s | %s |' % (r['test'], r['fixture'].replace('g5_sample_', ''), r['step'],
                                             ' | '.join('%.1e' % e for e in r['err']), ' | '.join('%.1e' % f for f in r['floor']),
                                             ' | '.join('%.2f' % x for x in ratio)))
print(f'\n{len(rows)} steps, {3 * len(rows)} (step, output) pairs; largest ratio {worst:.2f}; pairs above 3 x floor (admitted only by the 5 x TOL base): {over}')
