for v in '{"respawn_cooldown": 40}' '{"map_kwargs": {"box": 18.0, "exit_length": 70.0}}' '{"respawn_cooldown": 40, "map_kwargs": {"box": 18.0, "exit_length": 70.0}}'; do
  echo "### env_config=$v"
  python scripts/train_curve.py --stop 1500000 --every 150 --env-config "$v" 2>/dev/null | grep -v "^#"
done
echo "### N=30 default map"
python scripts/train_curve.py --stop 1500000 --every 150 --num-agents 30 2>/dev/null | grep -v "^#"
