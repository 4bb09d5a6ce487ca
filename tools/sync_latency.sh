#!/bin/bash
echo "== default"; python tools/sync_latency.py
echo "== ROC_ACTIVE_WAIT_TIMEOUT=1000"; ROC_ACTIVE_WAIT_TIMEOUT=1000 python tools/sync_latency.py
echo "== HSA_ENABLE_INTERRUPT=0"; HSA_ENABLE_INTERRUPT=0 python tools/sync_latency.py
echo "== CZ_GRAPHS=0"; CZ_GRAPHS=0 python tools/sync_latency.py
echo "== CZ_GRAPHS=0 HSA_ENABLE_INTERRUPT=0"; CZ_GRAPHS=0 HSA_ENABLE_INTERRUPT=0 python tools/sync_latency.py
echo "== DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/sync_latency.py
