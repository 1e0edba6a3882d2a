#!/bin/bash
# round-end evidence in one gpurun call: profile set of the hot step, then the widened rows
bash tools/collect_profiles.sh; echo "collect rc=$?"
bash tools/collect_widened.sh; echo "wide rc=$?"
