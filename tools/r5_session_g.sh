#!/bin/bash
# round-5 session G: the whole GPU suite, smoke, the driver's bench command, rocprofv3 kernel stats + the four PMC groups
bash tools/gpu_session.sh r5g tests smoke bench prof pmc
