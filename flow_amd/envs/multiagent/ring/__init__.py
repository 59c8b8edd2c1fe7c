"""Multi-agent environments on closed loops (flow/envs/multiagent/ring/)."""
