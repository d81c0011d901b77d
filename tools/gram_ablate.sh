for cfg in "4096 8 0" "16384 8 0"; do
 for a in 0 1 2 3; do
  GPX_GRAM_TPW=1 GPX_GRAM_ABLATE=$a timeout -k 10 200 python3 tools/gram_time.py $cfg | sed "s/^/ablate=$a  /" | cut -c1-90
 done
done
