# a bare `pytest` from the repo root must not collect the measurement one-offs under scripts/ (they run GPU work at import)
collect_ignore = ['scripts', 'gpurun_out', 'profiles']
