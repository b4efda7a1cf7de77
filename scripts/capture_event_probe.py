"""which event patterns does a hipGraph capture (torch.cuda.graph) survive?  each case in its own process (a failure is a segfault)"""
import sys, subprocess, os
CASES = ['one_event', 'many_events', 'wait_on_own_stream', 'unwaited_event', 'wait_twice', 'wait_after_join', 'drop_in_capture']
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
        print(f'{c:22s} rc={r.returncode} {r.stdout.strip()[-80:]} {r.stderr.strip()[-120:] if r.returncode else ""}')
    sys.exit(0)
import torch
case = sys.argv[1]
x = torch.zeros(1 << 20, device='cuda'); y = torch.zeros(1 << 20, device='cuda')
main = torch.cuda.Stream(); side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.stream(main):
    g.capture_begin()
    x += 1
    side.wait_stream(main)
    evs = []
    with torch.cuda.stream(side):
        for i in range(1 if case == 'one_event' else 6):
            y += 1
            e = torch.cuda.Event(); e.record(side); evs.append(e)
            if case == 'wait_on_own_stream':
                side.wait_event(e)
    if case == 'wait_after_join':
        main.wait_stream(side)
    if case != 'unwaited_event':
        for e in (evs if case != 'unwaited_event' else evs[:1]):
            main.wait_event(e)
            if case == 'wait_twice':
                main.wait_event(e)
    else:
        main.wait_event(evs[0])
    if case == 'drop_in_capture':
        del evs, e
    x += y
    main.wait_stream(side)
    g.capture_end()
torch.cuda.synchronize()
g.replay(); g.replay()
torch.cuda.synchronize()
print('ok', float(x[0]), float(y[0]))
