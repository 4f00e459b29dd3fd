#!/bin/bash
# round-5 session M (final tree): the whole GPU suite, smoke, the driver's bench command, rocprofv3 kernel stats + the four PMC groups, solver phase cycles
bash tools/gpu_session.sh r5m tests smoke bench prof pmc
