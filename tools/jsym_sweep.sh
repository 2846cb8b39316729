for c in 4 2 1; do echo "CPT $c"; QCDFT_JSYM8_CPT=$c timeout -k 10 200 python tools/jsym_time.py 2>&1 | grep "symmetric=2" ; done
echo auto; timeout -k 10 200 python tools/jsym_time.py 2>&1 | grep "symmetric=2"
