"""Old package path flow.envs.loop (now flow.envs.ring)."""
