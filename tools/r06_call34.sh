#!/bin/bash
# host cost of one launch on the ordinary dispatch path by argument-block size, and of a graph node with and without captured packets
mkdir -p gpurun_out
{ echo "== runtime default (captured graph packets)"; timeout 120 tools/micro/launch_host_cost
  echo "== DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 timeout 120 tools/micro/launch_host_cost; } 2>&1 | tee gpurun_out/r06_launch_host_cost.txt
