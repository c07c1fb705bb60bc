
template Program() {
   signal input GiCauiqG;
   signal input qmLmSHkX;
   signal input WFXGVEOP;
   signal input CGgipBry;
   signal output AnVdcrPq;
   AnVdcrPq<==((qmLmSHkX*(WFXGVEOP-WFXGVEOP))+(WFXGVEOP+GiCauiqG));
   for (var naqWwtHy = 14; (naqWwtHy<=16); naqWwtHy=(naqWwtHy+1)) {
      var BfRTXCJu = (6088816274827019151+1371361806776772192);
   }
   signal output SPxmhNOY;
   SPxmhNOY<==(((17422305444781129862+AnVdcrPq)*(AnVdcrPq+AnVdcrPq))+574449974085240956);
   for (var roqjRPZI = 12; (roqjRPZI<=56); roqjRPZI=(roqjRPZI+1)) {
      for (var kbbAtThA = 6; (kbbAtThA<=59); kbbAtThA=(kbbAtThA+1)) {
         signal output vOjPptSB;
         vOjPptSB<==(((16221837806467001880+SPxmhNOY)*(CGgipBry-10035514987322472385))+qmLmSHkX);
      }
   }
   signal output gsRtAttN;
   gsRtAttN<==(((qmLmSHkX+4986238944097284848)*(GiCauiqG-WFXGVEOP))+(qmLmSHkX-SPxmhNOY));
   for (var uVlYIlhA = 13; (uVlYIlhA<=59); uVlYIlhA=(uVlYIlhA+1)) {
      var iRubRsOZ = (10091321690987848788+2949550513007022802);
      iRubRsOZ=qmLmSHkX;
      uVlYIlhA=(9892124668431779745+15571534970633953339);
   }
   for (var fGPMXGtY = 2; (fGPMXGtY<=14); fGPMXGtY=(fGPMXGtY+1)) {
      signal output zuutHVPs;
      zuutHVPs<==(((qmLmSHkX-5987032064910053744)*(11499353099092102986-17235766849611554932))+(qmLmSHkX-qmLmSHkX));
      fGPMXGtY=(4874458250826207652+4863145321921129669);
      fGPMXGtY=(15714656421048359559*3344858387554254044);
      for (var wDNhwVCE = 24; (wDNhwVCE<=32); wDNhwVCE=(wDNhwVCE+1)) {
         var RqXjcNki[4];
         RqXjcNki[1]=(16231958633258309263*1490748384293732350);
         signal output acjpWnjo;
         acjpWnjo<==((CGgipBry*(zuutHVPs-13316479340215416473))+zuutHVPs);
         var KDIjWbmY = GiCauiqG;
      }
      signal output ggfJdsvg;
      ggfJdsvg<==(((SPxmhNOY+17632790591163495763)*AnVdcrPq)+CGgipBry);
   }
   var zGQmrEkT = CGgipBry;
   zGQmrEkT=(12852762995631484498*135974504023115972);
   signal output yQfgmNQe;
   yQfgmNQe<==(((GiCauiqG+CGgipBry)*(WFXGVEOP+qmLmSHkX))+(SPxmhNOY+GiCauiqG));
   signal output UYLXyyRF;
   UYLXyyRF<==(((9807163605918937193-gsRtAttN)*(WFXGVEOP-qmLmSHkX))+(gsRtAttN+WFXGVEOP));
   var uSPvneIp[5];
   zGQmrEkT=686409059227252890;
   zGQmrEkT=qmLmSHkX;
   uSPvneIp[4]=5384203161904868259;
   signal output ertqohfW;
   ertqohfW<==(((gsRtAttN+UYLXyyRF)*(CGgipBry-789831041917619177))+(GiCauiqG+UYLXyyRF));
   zGQmrEkT=GiCauiqG;
   signal output zVzaeRNU;
   zVzaeRNU<==(((qmLmSHkX+CGgipBry)*(2850982197869137082-10327639629911229682))+13861598008312513240);
   zGQmrEkT=zGQmrEkT;
   uSPvneIp[3]=2176226444334693414;
   for (var PbwFQDuz = 30; (PbwFQDuz<=34); PbwFQDuz=(PbwFQDuz+1)) {
      for (var LftzFXiW = 16; (LftzFXiW<=59); LftzFXiW=(LftzFXiW+1)) {
         var ERreZiTt[3];
         LftzFXiW=2302876001945739583;
      }
      uSPvneIp[0]=(2826868108607934043-14167522726082790020);
      var vazCDnfC[8];
      var xHgogvtX = (7211964529664080064+1545217765927158496);
   }
   zGQmrEkT=(17609169404144779825*15633827628725881807);
   uSPvneIp[2]=(uSPvneIp[0]);
   for (var jiBIybNT = 8; (jiBIybNT<=10); jiBIybNT=(jiBIybNT+1)) {
      uSPvneIp[4]=(uSPvneIp[4]);
      var VaxmMbpx[8];
      VaxmMbpx[2]=(1935057866427661365+1434901449229989353);
      zGQmrEkT=GiCauiqG;
      uSPvneIp[1]=(6371002114433633905-7364460533068507548);
   }
   signal output CnhboUHh;
   CnhboUHh<==(((17912695083168997899-10866579197248962212)*10989607563155254077)+yQfgmNQe);
   uSPvneIp[3]=CGgipBry;
   var ZIQDzZHw[5];
   uSPvneIp[4]=(uSPvneIp[4]);
   uSPvneIp[2]=zVzaeRNU;
   var XjXaAwqd = (9636797358723015414-3842167641110560084);
   uSPvneIp[4]=(934184829637608689*17150817601438862899);
   for (var EaPYkCUI = 3; (EaPYkCUI<=60); EaPYkCUI=(EaPYkCUI+1)) {
      for (var oAyFdmos = 9; (oAyFdmos<=43); oAyFdmos=(oAyFdmos+1)) {
         for (var dtMaEXGA = 14; (dtMaEXGA<=22); dtMaEXGA=(dtMaEXGA+1)) {
            for (var lyhTNDEl = 3; (lyhTNDEl<=9); lyhTNDEl=(lyhTNDEl+1)) {
               for (var pwrldrFF = 1; (pwrldrFF<=4); pwrldrFF=(pwrldrFF+1)) {
                  pwrldrFF=(uSPvneIp[4]);
               }
               for (var twCVTPTz = 4; (twCVTPTz<=23); twCVTPTz=(twCVTPTz+1)) {
                  var HCntGGJI = (uSPvneIp[1]);
               }
               var tZghxdqf = GiCauiqG;
            }
         }
      }
   }
   zGQmrEkT=(6648818653109980376+16097320490351526235);
   ZIQDzZHw[3]=(6544212417130763942-24859671027092271);
   for (var XhPouCKy = 0; (XhPouCKy<=54); XhPouCKy=(XhPouCKy+1)) {
      uSPvneIp[1]=(7161494443512528590*4447963146948259357);
      XjXaAwqd=(198266245510366807*13243015622125103486);
      signal output SquZPYgS;
      SquZPYgS<==((qmLmSHkX*15763042398867629835)+(gsRtAttN+ertqohfW));
      signal output eldjGGCN;
      eldjGGCN<==(((7450162553520842374-SPxmhNOY)*CnhboUHh)+(10362398225203997311+2556476037928866752));
   }
   signal output LYnwPGLB;
   LYnwPGLB<==(((7189620731031366757-gsRtAttN)*(qmLmSHkX+SPxmhNOY))+10194974211649809278);
   var FUlqiLUZ = (ZIQDzZHw[3]);
   signal output khAOCRNe;
   khAOCRNe<==((15990827632712448867*(CnhboUHh-gsRtAttN))+(UYLXyyRF+SPxmhNOY));
   signal output jzkpElOY;
   jzkpElOY<==(((ertqohfW+CnhboUHh)*gsRtAttN)+(khAOCRNe+UYLXyyRF));
   for (var ZvgxmxGP = 29; (ZvgxmxGP<=32); ZvgxmxGP=(ZvgxmxGP+1)) {
      var dvUOXsRP[9];
   }
   uSPvneIp[2]=(uSPvneIp[3]);
   var fpgltYVz = 17006444595140823980;
   for (var RrUlegpe = 19; (RrUlegpe<=29); RrUlegpe=(RrUlegpe+1)) {
      var OTUZWXMn[7];
      fpgltYVz=(8035912349311040723*8457800327716303475);
   }
   signal output AETRqHtR;
   AETRqHtR<==(((LYnwPGLB+18319914278602405759)*(gsRtAttN-13486993253046559322))+(13423208258620555623+jzkpElOY));
   ZIQDzZHw[4]=(9651288281199018314+931170307596627391);
   for (var OUWJXmgV = 17; (OUWJXmgV<=26); OUWJXmgV=(OUWJXmgV+1)) {
      var HfavptMG = (11014522557579346227-13273006930171386533);
      fpgltYVz=(2467409691970821617-12415103374874758797);
      uSPvneIp[2]=(16397943743176094920*9077174686082074344);
      signal output JPsoqIpN;
      JPsoqIpN<==(((yQfgmNQe-yQfgmNQe)*(AETRqHtR+16548239524799703617))+2446058916259299077);
   }
   signal output UaXTkLHB;
   UaXTkLHB<==(((2202369301854333119+UYLXyyRF)*(9727344792077072838-10382541927668856357))+gsRtAttN);
   XjXaAwqd=14472946135446725175;
   var uaFDwRwx[2];
   var IIThvhyz = (15474342259717155814*5904698496223496440);
   FUlqiLUZ=(15331029207314411434-1859618625681532638);
   for (var PKnotfSh = 24; (PKnotfSh<=29); PKnotfSh=(PKnotfSh+1)) {
      uaFDwRwx[1]=9451794738152627592;
      for (var MbHpUdvL = 4; (MbHpUdvL<=38); MbHpUdvL=(MbHpUdvL+1)) {
         signal output uOOJYaXE;
         uOOJYaXE<==((LYnwPGLB*(khAOCRNe+11299108045645779530))+(khAOCRNe+CGgipBry));
      }
      PKnotfSh=(14533813927818588790-6736752129800368881);
   }
   ZIQDzZHw[2]=UaXTkLHB;
   uSPvneIp[4]=17789384944784988205;
   var CUyPbxzU[7];
   for (var hXqLxVLQ = 3; (hXqLxVLQ<=30); hXqLxVLQ=(hXqLxVLQ+1)) {
      uSPvneIp[1]=SPxmhNOY;
   }
   for (var vWBFjLNQ = 16; (vWBFjLNQ<=54); vWBFjLNQ=(vWBFjLNQ+1)) {
      for (var dkmwqwqF = 24; (dkmwqwqF<=35); dkmwqwqF=(dkmwqwqF+1)) {
         ZIQDzZHw[4]=(uSPvneIp[0]);
         signal output KYDImNWP;
         KYDImNWP<==(((LYnwPGLB-ertqohfW)*(LYnwPGLB+qmLmSHkX))+(SPxmhNOY+8157620309934829640));
      }
      CUyPbxzU[5]=(uSPvneIp[1]);
      signal output CUeLUbbT;
      CUeLUbbT<==(((6479651978974089962-ertqohfW)*GiCauiqG)+zVzaeRNU);
      signal output yUbWKEfb;
      yUbWKEfb<==(((yQfgmNQe+GiCauiqG)*(17269764455676145520+ertqohfW))+jzkpElOY);
   }
   uSPvneIp[2]=(6820729840603100000*14922228941297176912);
   var DkonoISN[8];
   for (var OhGCyxim = 20; (OhGCyxim<=60); OhGCyxim=(OhGCyxim+1)) {
      signal output MoReUfAr;
      MoReUfAr<==((SPxmhNOY*(gsRtAttN+AnVdcrPq))+(CnhboUHh+yQfgmNQe));
   }
}

component main = Program();
        