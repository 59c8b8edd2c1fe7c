"""Old package path flow.multiagent_envs.loop (now flow.envs.multiagent.ring)."""
