function z = Aty(y,p,q)
% Drop-in shim with the reference's signature: forwards to libipdamg (HIP, gfx950)
% through the MEX gateway ipd_mex.  See INTEGRATION.md.
[z] = ipd_mex('Aty', y,p,q);
end
